#!/bin/bash
R=$GRAFT_REPO_ROOT; export MF_HIP_LIB=$R/recommender-system_amd/csrc/libmatfact_hip_exp.so
echo "== mid384 db nch48"; MF_SWEEP_MID=384 MF_SWEEP_MID_NCH=48 TRACE_OUT=$R/gpurun_out/trace_mid_db.txt bash tools/trace_skew.sh | tail -8
echo "== mid400 coop nch8"; MF_SWEEP_MID_KERNEL=coop MF_SWEEP_MID=400 MF_SWEEP_MID_NCH=8 TRACE_OUT=$R/gpurun_out/trace_mid_coop.txt bash tools/trace_skew.sh | tail -8
echo "== long2500 coop mid256"; MF_SWEEP_MID_KERNEL=coop MF_SWEEP_LONG=2500 MF_SWEEP_MID=256 MF_SWEEP_MID_NCH=13 TRACE_OUT=$R/gpurun_out/trace_mid_coop2.txt bash tools/trace_skew.sh | tail -8
echo "== noskew"; MF_SWEEP_SKEW=0 TRACE_OUT=$R/gpurun_out/trace_noskew.txt bash tools/trace_skew.sh | tail -4
