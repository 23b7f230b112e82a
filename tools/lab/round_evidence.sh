#!/bin/bash
# tools/lab/round_evidence.sh: everything the round's profiles/ files come from, on the library as it is, in the order that lets
# bench.py quote the traffic measured on THIS library -- the PMC passes first, copied into profiles/ on the box, then the bench lines.
#   gpurun: bash tools/lab/round_evidence.sh r06      (then, in the build container: tools/lab/collect_profiles.sh r06)
r=${1:-r06}
cd $GRAFT_REPO_ROOT
bash tools/lab/final_profiles.sh abc > gpurun_out/final_abc.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile.sh small_n89 89 300 1 2 > gpurun_out/prof_small_n89.log 2>&1
for d in gpurun_out/prof_*/; do tag=$(basename $d | sed 's/^prof_//'); [ -f $d/traffic.json ] && grep -q "$(sha256sum pgmuvi_amd/libpgmuvi_hip.so | cut -c1-16)" $d/traffic.json && cp $d/traffic.json profiles/${r}_pmc_hbm_traffic_${tag}.json; done
bash tools/lab/side_evidence.sh > gpurun_out/side.log 2>&1
bash tools/lab/tl_sizes.sh "256 512 1024" 4 1 > gpurun_out/tls.log 2>&1 && bash tools/lab/tl_sizes.sh "250" 3 2 >> gpurun_out/tls.log 2>&1
python3 tools/lab/soak.py > gpurun_out/soak.txt 2>&1
( echo "# PGM_FUZZ_CASES=120 PGM_FUZZ_SEED=7 python -m pytest tests/test_gpu_fuzz.py -m gpu -s -k random_light_curves   (library sha $(sha256sum pgmuvi_amd/libpgmuvi_hip.so | cut -c1-16))"
  PGM_FUZZ_CASES=120 PGM_FUZZ_SEED=7 python -m pytest tests/test_gpu_fuzz.py -m gpu -s -q -k random_light_curves 2>&1 | grep -E "^case|worst|passed|failed" ) > gpurun_out/fuzz_parity.txt
( echo "# PGM_FUZZ_RAGGED_CASES=40 PGM_FUZZ_SEED=7 python -m pytest tests/test_gpu_fuzz.py -m gpu -s -k ragged"
  PGM_FUZZ_RAGGED_CASES=40 PGM_FUZZ_SEED=7 python -m pytest tests/test_gpu_fuzz.py -m gpu -s -q -k ragged 2>&1 | grep -E "^ragged case|worst|passed|failed" ) > gpurun_out/fuzz_ragged.txt
( echo "# python tools/nutsconv.py on library sha $(sha256sum pgmuvi_amd/libpgmuvi_hip.so | cut -c1-16)"; timeout -k 10 600 python3 tools/nutsconv.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/nutsconv.txt
( echo "# PGM_FUZZ_CASES=300 PGM_FUZZ_SEED=23 python -m pytest tests/test_gpu_fuzz.py -m gpu -s -k random_light_curves   (library sha $(sha256sum pgmuvi_amd/libpgmuvi_hip.so | cut -c1-16))"
  PGM_FUZZ_CASES=300 PGM_FUZZ_SEED=23 timeout -k 10 500 python -m pytest tests/test_gpu_fuzz.py -m gpu -s -q -k random_light_curves 2>&1 | grep -E "^case|worst|passed|failed" ) > gpurun_out/fuzz300.txt
( echo "# PGM_FUZZ_RAGGED_CASES=120 PGM_FUZZ_SEED=31 python -m pytest tests/test_gpu_fuzz.py -m gpu -s -k ragged   (library sha $(sha256sum pgmuvi_amd/libpgmuvi_hip.so | cut -c1-16); call 17 of this seed found the last-bit inconsistency of round 6)"
  PGM_FUZZ_RAGGED_CASES=120 PGM_FUZZ_SEED=31 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -s -q -k ragged 2>&1 | grep -E "^ragged case|passed|failed" ) > gpurun_out/fuzz_ragged_120.txt
( echo "# PGM_FUZZ_SEED=77 PGM_FUZZ_SWITCH_CASES=3000 PGM_FUZZ_SMALL_CASES=20000 python -m pytest tests/test_gpu_fuzz.py -m gpu -s -k 'plainest or one_launch_value'   (library sha $(sha256sum pgmuvi_amd/libpgmuvi_hip.so | cut -c1-16))"
  PGM_FUZZ_SEED=77 PGM_FUZZ_SWITCH_CASES=3000 PGM_FUZZ_SMALL_CASES=20000 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -s -q -k "plainest or one_launch_value" 2>&1 | grep -E "light curves:|passed|failed" ) > gpurun_out/fuzz_bits_wide.txt
( echo "# python tools/lab/random_soak.py 4000 8   (library sha $(sha256sum pgmuvi_amd/libpgmuvi_hip.so | cut -c1-16))"; timeout -k 10 600 python3 tools/lab/random_soak.py 4000 8 2>&1 | grep -v "amdgpu\|Warning\|warn\|ystd" ) > gpurun_out/random_soak.txt
tail -1 gpurun_out/soak.txt; tail -2 gpurun_out/fuzz_parity.txt; tail -1 gpurun_out/fuzz_ragged.txt
python3 -c "
import json; d=json.load(open('gpurun_out/side/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['reference_published_workload']['train_native']['it_per_s'])"
