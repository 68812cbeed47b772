"""One-line view of a bench.py JSON line: `python scripts/show_bench.py bench.log` (the file is read; nothing from stdin)."""
import json
import sys

if len(sys.argv) < 2:
    sys.exit("usage: show_bench.py <file with a bench.py JSON line>")
try:
    lines = [l for l in open(sys.argv[1]) if l.startswith('{')]
except OSError as e:
    sys.exit("show_bench.py: %s" % e)
if not lines:
    sys.exit("show_bench.py: no JSON line in %s" % sys.argv[1])
d = json.loads(lines[-1])
print(sys.argv[1], d['value'], d['ms_per_step'], str(d.get('launch'))[:20], d.get('launch_probe'),
      d.get('kernels', {}).get('fwd_ms'), d.get('kernels', {}).get('bwd_ms'))
