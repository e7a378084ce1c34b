for b in 2 4 8 16; do for v in base side40 side80 side160; do
  if [ $v == base ]; then unset SIFTGPU_LIB; else export SIFTGPU_LIB=$PWD/build/variants/lib_$v.so; fi
  echo -n "B=$b $v: "; python3 profiles/frame_alone_trace.py run 1920 1080 60 $b
done; done
