export TMPDIR=/tmp
rm -rf data
( time python bench.py > /tmp/b.json 2> /tmp/b.err ) 2>&1 | tail -3
wc -l /tmp/b.json; tail -3 /tmp/b.err; python -c "
import json; d=json.loads(open('/tmp/b.json').read()); print(d['value'], d['extra']['render']['cpu_baseline'])"
