export SLAMPP_HIP_PLAN_TIMING=1
timeout 300 python tools/time_ba.py 2000 2000000 band > gpurun_out/c5_analyze.txt 2> gpurun_out/c5_analyze.err
timeout 300 python tools/time_ba.py 1000 500000 venice > gpurun_out/venice_analyze.txt 2> gpurun_out/venice_analyze.err
