cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
AB=$PWD/tactilesimulation_amd/csrc/ab
for tag in base u1 zeros; do
  if [ $tag = base ]; then unset TSIM_HIP_LIB; else export TSIM_HIP_LIB=$AB/libtsim_$tag.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ro_$tag -o ro -- python tools/readout_ab.py > gpurun_out/ro_$tag.log 2>&1
  echo "== readout $tag"; grep dtype gpurun_out/ro_$tag.log
  f=$(find gpurun_out/ro_$tag -name "*kernel_stats.csv" | head -1); grep -E "k_taxels|k_readout" $f | cut -c1-200
done
unset TSIM_HIP_LIB
echo "== k_forward shapes"
for cfg in "base 0" "base 32" "w2 16" "w2 32"; do set -- $cfg
  if [ $1 = base ]; then unset TSIM_HIP_LIB; else export TSIM_HIP_LIB=$AB/libtsim_$1.so; fi
  if [ $2 = 0 ]; then unset TSIM_LPE; else export TSIM_LPE=$2; fi
  echo "-- $1 LPE=$2"; for i in 1 2; do timeout 200 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc 2>/dev/null | tail -1; done
done
unset TSIM_HIP_LIB TSIM_LPE
echo "== literal tests"
timeout 600 python -m pytest tests/test_gpu_literal.py -q -x -s 2>&1 | grep -vE "^$|Warning|warn" | tail -25
