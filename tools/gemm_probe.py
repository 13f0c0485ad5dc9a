"""Times the lstm_xproj_x3 launch group under the KRK_X3_DBG ablation bits (run once per value)."""
import os, sys, json, subprocess
for spec in sys.argv[2:]:
    dbg = spec
    env = dict(os.environ, KRK_X3_DBG=dbg)
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '24', '--warmup', '4', '--slots', '1', '--no-cpu-baseline'],
                         env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    print('dbg', spec, [(l['name'], l['ms']) for l in d['launches'] if any(k in l['name'] for k in sys.argv[1].split(','))], flush=True)
