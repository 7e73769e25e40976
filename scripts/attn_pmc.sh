#!/bin/bash
# PMC passes over the attention kernel alone (one shape, SHAPE/VAR env as scripts/attn_only.py): gpurun_out/<tag>_pmc.json
set -e -o pipefail
cd "$(dirname "$0")/.."
TAG=${1:-attnpmc}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
P3="SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $P --kernel-include-regex pgn_attn --output-format csv -d $OUT/${TAG}_p$i -- python3 scripts/attn_only.py > /dev/null 2> $OUT/${TAG}_p$i.err
  echo "pass $i done"
done
python3 scripts/profile_summary.py pmc $OUT/${TAG}_pmc.json $OUT/${TAG}_p1 $OUT/${TAG}_p2 $OUT/${TAG}_p3 > /dev/null
rm -rf $OUT/${TAG}_p1 $OUT/${TAG}_p2 $OUT/${TAG}_p3
