#!/bin/bash
# SQ counters of the attention kernel (tools/attn_time.py's launches): two rocprofv3 --pmc passes, summarised per kernel / grid.
R=${GRAFT_REPO_ROOT:-$(pwd)}; case "$COMORAG_HIP_LIB" in /*|"") ;; *) export COMORAG_HIP_LIB=$R/$COMORAG_HIP_LIB;; esac; O=$R/gpurun_out/${1:-attn_pmc}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace -d $O/a -o w -- python $R/tools/attn_time.py > /dev/null 2> $O/a.err
python $R/tools/rocpd_pmc.py $O/a/w_results.db attn_fwd 10 > $O/attn_pmc_a.jsonl
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace -d $O/b -o w -- python $R/tools/attn_time.py > /dev/null 2> $O/b.err
python $R/tools/rocpd_pmc.py $O/b/w_results.db attn_fwd 10 > $O/attn_pmc_b.jsonl
rm -rf $O/a $O/b
python - <<PY
import json
for f in ("$O/attn_pmc_a.jsonl", "$O/attn_pmc_b.jsonl"):
    for l in open(f):
        d = json.loads(l)
        print(d["kernel"][-40:], d["grid"], d["counter"], f"{d['avg']:.4g}", f"{d['avg_kernel_us']:.1f}us", d["launches"])
PY
