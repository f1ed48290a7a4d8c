#!/bin/bash
# PC sampling of the call kernel (debug-line variant of the engine): tools/pcs_run.sh <tag>
# Writes gpurun_out/pcs_<tag>_{host_trap,stochastic}.json (histograms by instruction / source line).
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export VLR_LIB=$R/varlociraptor_amd/matrix/libvlr_dbg.so
mkdir -p gpurun_out
for m in host_trap stochastic; do
  rm -rf /tmp/pcs_$m
  if [ $m = host_trap ]; then U="--pc-sampling-unit time --pc-sampling-interval 2000"; else U="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; fi
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $m $U -d /tmp/pcs_$m -o pcs --output-format csv -- python tools/rate_variant.py > gpurun_out/pcs_${tag}_$m.log 2>&1
  echo "rc $m $?" >> gpurun_out/pcs_${tag}_$m.log
  ls -la /tmp/pcs_$m/* >> gpurun_out/pcs_${tag}_$m.log 2>&1
  python tools/pcs_aggregate.py /tmp/pcs_$m gpurun_out/pcs_${tag}_$m.json >> gpurun_out/pcs_${tag}_$m.log 2>&1
done
tail -5 gpurun_out/pcs_${tag}_*.log
