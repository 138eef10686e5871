import sys, json, subprocess
for n in (512, 1024, 2048, 4096):
    for mode, name in ((16, 'sync'), (8, 'async')):
        import os
        env = dict(os.environ, KSFD_TUNE=str(1 | mode))
        out = subprocess.run([sys.executable, 'bench.py', '--grid', str(n), '--no-cpu-baseline', '--steps', '5', '--warmup', '2'], capture_output=True, text=True, env=env).stdout
        line = [l for l in out.splitlines() if l.startswith('{')]
        if not line:
            print(n, name, 'FAILED', out[-300:]); continue
        d = json.loads(line[-1])
        print('n %5d %-5s ms/step %8.3f its/step %.1f value %.3e' % (n, name, d['ms_per_step'], d['config']['linear_its_per_step'], d['value']), flush=True)
