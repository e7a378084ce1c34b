set -e
cd sift_amd/host
for lanes in 1 2 3 4; do for sb in 8 16; do for th in 16 32; do
  echo -n "lanes=$lanes shared=$sb threads=$th  "
  SIFT_GPU_LANES=$lanes timeout -k 10 120 ./sift_host_bench 1920 1080 $th 20 0 $sb | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['images_per_s'], d['sift_ms_single'])"
done; done; done
