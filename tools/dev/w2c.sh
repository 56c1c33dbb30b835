#!/bin/bash
# Developer aid: compile csrc/fused_wide2.hip alone with the build's flags, print registers / scratch / SGPR spills per kernel and where
# (by MFMA count) the scratch operations of one instantiation sit.   bash tools/dev/w2c.sh [mangled-name fragment]
R=$(cd "$(dirname "$0")/../.." && pwd)
cd $R/camouflage_multimodal_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -fno-honor-nans -c fused_wide2.hip -o /tmp/fw2.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "error|warning: [^u]|Function Name|VGPRs:|Scratch|SGPRs Spill" | paste - - - - | sed 's/\[-Rpass[^]]*\]//g; s/remark: fused_wide2.hip:[0-9]*:[0-9]*://g; s/_ZN12_GLOBAL__N_1//' | cut -c1-190
S=/tmp/fused_wide2-hip-amdgcn-amd-amdhsa-gfx950.s
if [ -n "$1" ]; then
  start=$(grep -n "^_ZN12_GLOBAL__N_1.*$1" $S | head -1 | cut -d: -f1)
  end=$(awk -v s=$start 'NR>s && /^.Lfunc_end/{print NR; exit}' $S)
  awk -v s=$start -v e=$end 'NR>=s && NR<=e' $S | awk '/v_mfma/{m++} /scratch_/{c[m]++} END{for(k in c) print "  after "k" mfmas: "c[k]" scratch ops"}' | sort -t' ' -k4 -n
fi
