set -u
# The randomised campaigns on the library as it stands (GPU box, repo root): handle path + window kernels, the banded
# direct solver with closures, the sharded direct solver with closures. Logs in gpurun_out/fuzz/; summarised into
# profiles/rNN_fuzz_parity.txt by hand (the last line of every log is the campaign's own summary).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/fuzz
mkdir -p $OUT
for s in 701 702; do timeout 900 python tools/fuzz_parity.py --cases 2000 --seed $s > $OUT/fzp_$s.log 2>&1; tail -1 $OUT/fzp_$s.log | cut -c1-400; done
timeout 900 python tools/fuzz_parity.py --cases 1500 --seed 703 --device-build > $OUT/fzp_703.log 2>&1; tail -1 $OUT/fzp_703.log | cut -c1-400
timeout 900 python tools/fuzz_parity.py --cases 300 --seed 704 --large > $OUT/fzp_704.log 2>&1; tail -1 $OUT/fzp_704.log | cut -c1-400
IROTAVG_BAND_DIRECT=1 timeout 600 python tools/fuzz_parity.py --cases 1500 --seed 705 --nmax 60 > $OUT/fzp_705_forced_direct.log 2>&1; tail -1 $OUT/fzp_705_forced_direct.log | cut -c1-400
for s in 711 712 713; do timeout 900 python tools/fuzz_band_direct.py --seed $s --cases 1000 > $OUT/fzb_$s.log 2>&1; tail -1 $OUT/fzb_$s.log; done
for s in 721 722 723; do timeout 900 python tools/fuzz_sharded_direct.py --seed $s --cases 120 --closures-max 1000 > $OUT/fzs_$s.log 2>&1; tail -1 $OUT/fzs_$s.log; done
timeout 600 python -m pytest tests/test_gpu_band_direct.py -x -q -m gpu -k "round6" 2>&1 | grep -a "passed\|failed"
