#!/bin/bash
# PMC pass over one microbenchmark (counters in their own run: no --stats/--kernel-trace mix with sys traces)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/microbench.py all > gpurun_out/microbench.log 2>&1; cat gpurun_out/microbench.log | grep -v amdgpu.ids
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z_]+|TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+)\b" | sort -u > gpurun_out/counters.txt
wc -l gpurun_out/counters.txt
grep -E "MFMA|SQ_WAIT|SQ_BUSY_CY|BANK_CONFLICT|SQ_WAVE_CYCLES|ACTIVE_INST|GUI_ACTIVE|SQ_INSTS_VALU$|SQ_INSTS_LDS|INST_LEVEL" gpurun_out/counters.txt | tr '\n' ' '
WHAT=${1:-"conv block3.1"}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT -d "$OLDPWD/gpurun_out/pmc1" -o pmc --output-format csv -- python "$OLDPWD/tools/microbench.py" $WHAT --iters 5 > "$OLDPWD/gpurun_out/pmc1.log" 2>&1; echo rc=$?)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$OLDPWD/gpurun_out/pmc2" -o pmc --output-format csv -- python "$OLDPWD/tools/microbench.py" $WHAT --iters 5 > "$OLDPWD/gpurun_out/pmc2.log" 2>&1; echo rc=$?)
find gpurun_out/pmc1 gpurun_out/pmc2 -name "*.csv" | head
tail -3 gpurun_out/pmc1.log
