import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d['launch'][:20], d['launch_probe'], d.get('kernels', {}).get('fwd_ms'), d.get('kernels', {}).get('bwd_ms'))
