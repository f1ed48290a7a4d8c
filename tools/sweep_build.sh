#!/bin/bash
# Compiler-flag / source-toggle sweep of the call kernel (round 6): one library per variant, only vlr_kernels.hip recompiled
# (tools/dbg_variant.sh), eight compilations at a time.  tools/rate_variants.sh times them on the GPU box (libvlr_x_*.so).
cd "$(dirname "$0")/.."
v() { name=$1; shift; echo "$name|$*"; }
{
v x_memclause -mllvm -amdgpu-sched-strategy=max-memory-clause
v x_iterilp -mllvm -amdgpu-sched-strategy=iterative-ilp
v x_itermaxocc -mllvm -amdgpu-sched-strategy=iterative-maxocc
v x_iterminreg -mllvm -amdgpu-sched-strategy=iterative-minreg
v x_bias0 -mllvm -amdgpu-schedule-metric-bias=0
v x_bias100 -mllvm -amdgpu-schedule-metric-bias=100
v x_nohrp -mllvm -amdgpu-disable-unclustered-high-rp-reschedule
v x_noclo -mllvm -amdgpu-disable-clustered-low-occupancy-reschedule
v x_trackers -mllvm -amdgpu-use-amdgpu-trackers
v x_nofresh -DVLR_DBG_NO_FRESH_LANE
v x_nopostmi -mllvm -enable-post-misched=0
v x_prio1 -DVLR_PRIO=1
v x_prio2 -DVLR_PRIO=2
v x_ilpnofresh -mllvm -amdgpu-sched-strategy=max-ilp -DVLR_DBG_NO_FRESH_LANE
v x_relaxocc -mllvm -amdgpu-schedule-relaxed-occupancy
v x_reghold4 -DVLR_DBG_REGHELD=4
} | xargs -P 8 -I{} bash -c 'IFS="|" read name flags <<< "{}"; STRAT=default tools/dbg_variant.sh $name $flags > /tmp/sweep_$name.log 2>&1 || echo "FAILED $name"'
ls varlociraptor_amd/matrix/ | grep x_
