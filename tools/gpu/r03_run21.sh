cd $GRAFT_REPO_ROOT
time python __graft_entry__.py --smoke 2>&1 | tail -3
time python bench.py > /tmp/b.json 2>/tmp/b.err; python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['config4']['utterances_per_sec'], d['half_mode']['utterances_per_sec'])"
