# same-box A/B of two builds of the library on the whole train step: tools/bin/libsvs_hip_A.so against the tree's
set -o pipefail
for i in 1 2 3; do
SVS_LIB_PATH=tools/bin/libsvs_hip_A.so python bench.py --steps 40 --warmup 8 --no-extras --no-layers --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A', d['ms_per_step'])"
python bench.py --steps 40 --warmup 8 --no-extras --no-layers --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', d['ms_per_step'])"
done
