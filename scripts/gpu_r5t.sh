#!/bin/bash
# round 5, call T: phase stamps of the estimator's GEMM kernels
mkdir -p gpurun_out/r5t
cd /root/repo
timeout 300 ab_libs/est_phases > gpurun_out/r5t/est_phases.log 2>&1
cat gpurun_out/r5t/est_phases.log
