#!/bin/bash
# tools/lab/collect_round.sh r06: after `gpurun -- bash tools/lab/round_evidence.sh r06` -- copies this run's evidence into profiles/ and drops
# anything that was not measured on the library that is in the tree now (older profile directories linger in gpurun_out/).
r=${1:-r06}
bash tools/lab/collect_profiles.sh $r > /dev/null 2>&1
for d in gpurun_out/prof_*/; do tag=$(basename $d | sed 's/^prof_//'); if [ -z "$(find $d -maxdepth 1 -newer pgmuvi_amd/libpgmuvi_hip.so -name 'kernel_stats.csv' | head -1)" ]; then rm -f profiles/${r}_*_${tag}.csv profiles/${r}_*_${tag}.json profiles/${r}_*_${tag}.txt; fi; done
rm -f profiles/${r}_kernel_stats_small.csv profiles/${r}_pmc_hbm_traffic_ragged512.json profiles/${r}_pmc_mfma_util_ragged512.json
cp gpurun_out/prof_small/kernel_stats.csv profiles/${r}_kernel_stats_small_path.csv
cp gpurun_out/prof_small_n89/traffic.json profiles/${r}_pmc_hbm_traffic_small_n89.json
cp gpurun_out/prof_small_n89/mfma_util.json profiles/${r}_pmc_mfma_util_small_n89.json
cp gpurun_out/prof_small_n89/kernel_stats.csv profiles/${r}_kernel_stats_small_n89.csv
for f in soak fuzz_parity fuzz_ragged; do grep -v amdgpu.ids gpurun_out/$f.txt > profiles/${r}_$f.txt; done
[ -f gpurun_out/nutsconv.txt ] && grep -v amdgpu.ids gpurun_out/nutsconv.txt > profiles/${r}_nuts_config5_convergence.txt
[ -f gpurun_out/fuzz300.txt ] && grep -v amdgpu.ids gpurun_out/fuzz300.txt > profiles/${r}_fuzz_parity_300.txt
[ -f gpurun_out/fuzz_ragged_120.txt ] && grep -v amdgpu.ids gpurun_out/fuzz_ragged_120.txt > profiles/${r}_fuzz_ragged_120.txt
[ -f gpurun_out/random_soak.txt ] && grep -v amdgpu.ids gpurun_out/random_soak.txt > profiles/${r}_random_soak.txt
[ -f gpurun_out/fuzz_bits_wide.txt ] && grep -v amdgpu.ids gpurun_out/fuzz_bits_wide.txt > profiles/${r}_fuzz_bits_wide.txt
for d in gpurun_out/tls_*/; do [ -d "$d" ] || continue; k=$(find $d -name '*kernel_trace.csv' -newer pgmuvi_amd/libpgmuvi_hip.so | head -1); [ -n "$k" ] && python3 tools/timeline.py "$k" > profiles/${r}_timeline_$(basename $d | sed 's/^tls_//').txt; done
ls profiles | grep -c ${r}_; grep -h "_lib_sha16" profiles/${r}_pmc_hbm_traffic_*.json | sort | uniq -c; sha256sum pgmuvi_amd/libpgmuvi_hip.so | cut -c1-16
