#!/bin/bash
# kernel ms of config3 / config2 (200 k loci) for the shipped library and every tuning variant varlociraptor_amd/matrix/libvlr_x_*.so
python tools/rate_variant.py 2>/dev/null
for so in varlociraptor_amd/matrix/libvlr_x_*.so; do
  VLR_LIB=$PWD/$so python tools/rate_variant.py 2>/dev/null
done
