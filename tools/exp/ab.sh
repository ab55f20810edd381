mkdir -p gpurun_out/r02s
DIGAT_XATTN_STAGED=1 DIGAT_STAGED_CFG=4 DIGAT_STREAM_TIMERS=1 python tools/exp/timers.py > gpurun_out/r02s/timers.txt 2>&1
tail -n 14 gpurun_out/r02s/timers.txt
