# round 4, GPU session A: full GPU suite, the driver-shaped bench line, LPT A/B for episode launches, the SURVEY-worded insertion leg
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r04a
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > ${O}_tests.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 --pmc-dump gpurun_out/r04a_pmc_f32.json > ${O}_bench.json ) 2> ${O}_bench.err
for i in 1 2; do
  timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_lpt_on.jsonl 2>/dev/null
  TSIM_NO_EPISODE_LPT=1 timeout 120 python bench.py --steps 20 --warmup 5 --timed-only --no-pmc >> ${O}_lpt_off.jsonl 2>/dev/null
done
timeout 300 python bench.py --workload insertion --steps 18 --warmup 9 --no-pmc --no-cpu-baseline --no-closed-loop > ${O}_insertion.json 2> ${O}_insertion.err
timeout 300 python tools/insertion_attempt_probe.py > ${O}_insertion_probe.json 2> ${O}_insertion_probe.err
tail -3 ${O}_tests.log; head -c 600 ${O}_bench.json; echo; cat ${O}_lpt_on.jsonl ${O}_lpt_off.jsonl; head -c 400 ${O}_insertion.json; echo; cat ${O}_insertion_probe.json | head -c 1500
