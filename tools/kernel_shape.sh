#!/bin/bash
# size, branch count, registers and scratch of the device kernels in csrc/obj/<unit>.o whose mangled name matches a pattern
#   bash tools/kernel_shape.sh gemm 'gemm2_kernelIDF16bLi1[29][82]ELi128ELi4'
U=${1:-gemm}; PAT=${2:-gemm2_kernel}
L=/opt/rocm/lib/llvm/bin; D=/tmp/dis_$U; mkdir -p $D
ROOT=$(cd $(dirname $0)/.. && pwd)
$L/llvm-objcopy --dump-section=.hip_fatbin=$D/fat.bin $ROOT/ubisoft-laforge-msmd_amd/csrc/obj/$U.o $D/copy.o
$L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$D/fat.bin --output=$D/$U.co
$L/llvm-objdump -d $D/$U.co > $D/$U.s
$L/llvm-readelf --notes $D/$U.co > $D/$U.notes
python3 - $D/$U.s $D/$U.notes "$PAT" <<'PY'
import re, sys
lines = open(sys.argv[1]).read().split('\n'); pat = re.compile(sys.argv[3])
notes = open(sys.argv[2]).read()
regs = {}
name = None
for l in notes.splitlines():
    l = l.strip()
    if l.startswith('.name:'): name = l.split(':', 1)[1].strip(); regs[name] = {}
    elif name and (l.startswith('.vgpr_count:') or l.startswith('.private_segment_fixed_size:') or l.startswith('.sgpr_count:')): regs[name][l.split(':')[0][1:]] = int(l.split(':')[1])
starts = [(i, l) for i, l in enumerate(lines) if re.match(r'^[0-9a-f]+ <_Z', l)]
for k, (i, l) in enumerate(starts):
    nm = l.split('<')[1].rstrip('>:')
    if not pat.search(nm): continue
    j = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    body = lines[i:j]
    r = regs.get(nm, {})
    print(f"{nm[:100]:100s} lines {j - i:6d} branches {sum('s_cbranch' in b or 's_branch' in b for b in body):5d} vgpr {r.get('vgpr_count')} scratch {r.get('private_segment_fixed_size')}")
PY
