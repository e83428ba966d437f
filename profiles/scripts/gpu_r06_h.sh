#!/bin/bash
# round 6, call h: the whole GPU suite on HEAD (the stage-event test post-dates the r06_z evidence call) + the default bench line on a second box of the pool + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_h; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt >> $O/rc.txt
( time python bench.py ) > $O/bench_f16.json 2> $O/bench_f16.err; echo "bench rc=$?" >> $O/rc.txt
python bench.py --dtype f8 --no-extras --no-cpu-baseline > $O/bench_f8.json 2> $O/bench_f8.err; echo "bench f8 rc=$?" >> $O/rc.txt
cat $O/rc.txt; head -c 400 $O/bench_f16.json
