"""CPU oracle for the audio -> motion-coefficient hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain numpy (fp32) restatement
of the reference's algorithm (ubisoft/ubisoft-laforge-msmd) for the path named
in BASELINE.json.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker;
the product package (``ubisoft-laforge-msmd_amd/``) never imports it and fails
loudly when its HIP library is missing.

Pinning: the reference ships no tests, golden vectors or fixtures
(SURVEY.md §4), and part of the arithmetic lives in third-party code that is
not under /root/reference (transformers==4.44.2 wav2vec2/hubert modules,
torch==2.0.0 nn.TransformerDecoder/EncoderLayer).  The oracle
is therefore pinned against outputs of the reference itself, imported in the
build container with the shim of SURVEY.md Appendix C and run on closed-form
synthetic weights: ``tests/golden/make_goldens.py`` generated the committed
``tests/golden/*.npz`` vectors and ``tests/test_oracle_vs_golden.py`` checks
every oracle function against them (fp32 tolerance stated per test; index
tables bit-exact).

Every function cites the reference file:line it follows.
"""
