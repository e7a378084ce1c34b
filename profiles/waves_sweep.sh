for w in 0 2560 10240 18432 36864; do echo -n "ORI_WAVES=$w: "; SIFTGPU_ORI_WAVES=$w timeout -k 10 100 python3 profiles/frame_alone_trace.py run 1920 1080 200 || exit 1; done
for w in 2560 10240 20480 40960; do echo -n "DESCR_WAVES=$w: "; SIFTGPU_DESCR_WAVES=$w timeout -k 10 100 python3 profiles/frame_alone_trace.py run 1920 1080 200 || exit 1; done
