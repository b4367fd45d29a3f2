# round 6: the cleaned tree -- full GPU suite, the bench line, A/B against round 5's library
cd /root/repo; O=gpurun_out/r06f; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
python bench.py --steps 20 > $O/bench20.json 2> $O/bench20.err; python3 -c "
import json; j = json.load(open('$O/bench20.json')); print('bench20', j['value'], j['ms_per_step'], j['timed_blocks'], j['roofline']['frac'], j.get('logprob_mae'))"
