#!/bin/bash
# Runs `python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline` against build/variants/lib_<name>.so for every name
# given ("base" = the library in the tree) and prints throughput, step time and the stage times named in $STAGES.
STAGES=${STAGES:-detect}
for name in "$@"; do
  if [ "$name" == "base" ]; then unset SIFTGPU_LIB; else export SIFTGPU_LIB=$PWD/build/variants/lib_$name.so; fi
  timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > /tmp/vb.json 2>/tmp/vb.err || { echo "$name failed"; tail -2 /tmp/vb.err; continue; }
  python3 - "$name" "$STAGES" <<'P'
import json, sys
d = json.loads(open('/tmp/vb.json').read().strip().splitlines()[-1])
st = d['stage_ms_per_step']
print("%-12s %7.1f img/s  step %6.1f ms | %s | det %d GB/s  features %d" % (sys.argv[1], d['value'], d['ms_per_step'],
      " ".join("%s %.2f" % (k, st.get(k, 0)) for k in sys.argv[2].split(',')), d['detect_kernel']['GBps'], d['features_last_step']))
P
done
