set -e
TAG=${TAG:-r04_h}
bash tools/round_profiles.sh $TAG
b() { name=$1; shift; python bench.py --cpu-seconds 0 --no-pcg "$@" > gpurun_out/${TAG}_bench_$name.json 2> gpurun_out/${TAG}_bench_$name.err; python -c "import sys,json; d=json.loads(open('gpurun_out/${TAG}_bench_$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],2), round(d['ms_per_step'],2), d.get('schur_pattern',{}).get('camera_sequence'))"; }
b venice_locality013 --locality 0.13
b venice_locality013_shuffled --locality 0.13 --shuffle-cameras
b venice_plane008 --plane-radius 0.08
b final13682_f32_locality003 --workload final-13682 --locality 0.03 --facto-type f32 --steps 5 --warmup 1
b final13682_f32_locality003_shuffled --workload final-13682 --locality 0.03 --facto-type f32 --steps 5 --warmup 1 --shuffle-cameras
b final13682_f32_plane003 --workload final-13682 --plane-radius 0.03 --facto-type f32 --steps 5 --warmup 1
b dubrovnik --workload dubrovnik-356
b ladybug --workload ladybug-49
BA_SPARSE_S=1 python bench.py --cpu-seconds 0 --no-pcg --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('venice dense through the list schedule with look-ahead', d['value'], d['ms_per_step'])"
bash tools/trace_sparse.sh ${TAG}_final_two_runs --workload final-13682 --locality 0.03 --facto-type f32 --steps 3 --warmup 1
echo done
