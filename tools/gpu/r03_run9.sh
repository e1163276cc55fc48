bash tools/gpu/mrf_variants.sh "-DMRF_C=16 -DMRF_T=256 -DMRF_NW=4" "-DMRF_C=8 -DMRF_T=256 -DMRF_NW=4"
