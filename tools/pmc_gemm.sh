cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_gemm; mkdir -p $OUT
for shape in "6400 768 3072" "6400 2304 768" "16384 4096 3072"; do
  tag=$(echo $shape | tr ' ' 'x')
  i=0
  for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_${tag}_$i -o p -- python3 /root/repo/tools/one_gemm.py 17 $shape 1 > /tmp/pmc.log 2>&1
    f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp $f $OUT/${tag}_pass$i.csv
  done
done
ls $OUT
