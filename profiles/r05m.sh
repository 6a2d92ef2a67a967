#!/bin/bash
# r05m: stem_wgrad_nb_kernel - pixel-slice counts, and 3 workgroups per CU (168-register build) against 2
mkdir -p gpurun_out
(SPLITS=512,768,1024,1536,2048,3072 timeout 300 python profiles/stem_wgrad_bench.py; echo "--- -DLP_STEM_NB_WGS=3"; LP_HIP_LIB=$PWD/build/liblp_hip_nb3.so SPLITS=768,1024,1536,2048,3072 timeout 300 python profiles/stem_wgrad_bench.py) > gpurun_out/r05m_stem_wgrad_variants.txt 2>&1
cat gpurun_out/r05m_stem_wgrad_variants.txt
