#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "--- gemm4 no stagger"; FT=2256 python tools/bench_gemm_cold.py
for s in 1 2 4; do echo "--- gemm4 stagger $s"; SAM_GEMM4_STAGGER=$s FT=2256 python tools/bench_gemm_cold.py; done
