set -u
# second session of round 5: the campaigns on the final library, then tools/final_profiles.sh
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/fzs_*.log gpurun_out/fzb_*.log
timeout 200 python tools/fuzz_sharded_direct.py --seed 21 --cases 60 --closures-max 400 > gpurun_out/fzs_21.log 2>&1
for s in 22 31 32 33; do timeout 400 python tools/fuzz_sharded_direct.py --seed $s --cases 100 --closures-max 1000 > gpurun_out/fzs_$s.log 2>&1; done
for s in 61 71; do timeout 200 python tools/fuzz_band_direct.py --seed $s --cases 300 > gpurun_out/fzb_$s.log 2>&1; done
tail -1 gpurun_out/fzs_*.log gpurun_out/fzb_*.log
bash tools/final_profiles.sh
