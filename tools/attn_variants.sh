#!/bin/bash
# Kernel time of the attention kernel by build variant (rocprofv3 --kernel-trace: the kernel's own duration, not the call-to-call time of a Python loop)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" comorag_amd/lib/libatt_d1p0.so comorag_amd/lib/libatt_d0p1.so comorag_amd/lib/libattold.so; do   # default = swizzled layout through registers + packed softmax; d1p0 = LDS-DMA + packed; d0p1 = padded layout + packed; attold = round 5
  rm -rf /tmp/av; if [ -n "$lib" ]; then export COMORAG_HIP_LIB=$R/$lib; else unset COMORAG_HIP_LIB; fi
  rocprofv3 --kernel-trace -d /tmp/av -o w -- python $R/tools/attn_time.py > /dev/null 2>&1
  echo "== ${lib:-default}"; python $R/tools/rocpd_stats.py /tmp/av/w_results.db 2>/dev/null | grep attn_fwd | head -4
done; done
