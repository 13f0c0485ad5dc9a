import os, sys, subprocess, json
sys.path.insert(0, '.')
lib = os.path.abspath('kraken_amd/libkraken_amd_ablate.so')
for v in (4, 3):
    for dbg in (0, 1, 3, 7, 15, 31, 63, 4, 5, 12, 13, 29):
        e = dict(os.environ, KRK_LSTM_V=str(v), KRAKEN_AMD_LIB=lib, KRK_LSTM_DBG=str(dbg))
        out = subprocess.run([sys.executable, 'tools/lstm_ws_probe.py', 'child', '256', '150', ''], env=e, capture_output=True, text=True, timeout=120)
        try:
            r = json.loads(out.stdout.strip().splitlines()[-1]).get('lstm_rec_x3')
        except Exception:
            r = (out.stderr or out.stdout)[-300:]
        print('v', v, 'dbg', dbg, r, flush=True)
