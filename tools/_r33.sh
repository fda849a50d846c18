python -c "
import __graft_entry__ as g, time
t=time.time(); g.build(); print('build ok', round(time.time()-t,1))
t=time.time(); g.smoke(); print('smoke ok', round(time.time()-t,1))
"
python -m pytest tests/ -x -q -m gpu 2>&1 | grep "passed\|failed"
