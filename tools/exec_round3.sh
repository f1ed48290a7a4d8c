#!/bin/bash
# Third GPU run of round 6: the reproducer of the asm-statement defects, the GPU suite (build matrix with -O1 at four waves and max-ILP with
# fresh_lane back in it), kernel times of the shipped and the max-ILP build.
R=$PWD; M=$R/varlociraptor_amd/matrix; O=$R/gpurun_out/r06_exec3
mkdir -p $O
{
echo "== reproducer"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/repro/asm_lane_read_hazard.hip -o /tmp/asm_repro 2>/dev/null && /tmp/asm_repro
echo "== kernel time, 200 000 loci"
timeout 600 python tools/rate_variant.py
VLR_LIB=$M/libvlr_ilp.so timeout 600 python tools/rate_variant.py
echo "== GPU suite"
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
} 2>&1 | tee $O/log.txt
