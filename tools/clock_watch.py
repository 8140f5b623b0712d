"""Measurement aid: samples the GPU's clocks and power (rocm-smi / amd-smi sysfs files) while a bench workload runs in a loop.
usage: python tools/clock_watch.py <workload> [seconds]"""
import glob, os, subprocess, sys, time
workload = sys.argv[1]; seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
def read(path):
    try: return open(path).read().strip()
    except OSError: return None
cards = [c for c in glob.glob('/sys/class/drm/card*/device') if os.path.exists(os.path.join(c, 'pp_dpm_sclk'))]
hwmons = glob.glob(cards[0] + '/hwmon/hwmon*') if cards else []
def sample():
    out = {}
    if cards:
        for name in ('pp_dpm_sclk', 'pp_dpm_mclk', 'pp_dpm_fclk'):
            text = read(os.path.join(cards[0], name)) or ''
            active = [line for line in text.splitlines() if line.endswith('*')]
            out[name] = active[0] if active else text.replace('\n', ' | ')[:80]
        out['busy'] = read(os.path.join(cards[0], 'gpu_busy_percent'))
    for hw in hwmons:
        for name in ('power1_average', 'power1_input', 'freq1_input', 'freq2_input', 'temp1_input'):
            value = read(os.path.join(hw, name))
            if value is not None: out[name] = value
    return out
print('idle', sample(), flush=True)
env = dict(os.environ)
proc = subprocess.Popen([sys.executable, 'bench.py', '--workload', workload, '--no-cpu-baseline', '--no-extras', '--steps', str(int(seconds * 4000 if workload != 'cinematic' else seconds * 1200)), '--warmup', '100'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env)
t0 = time.time()
while proc.poll() is None:
    time.sleep(0.5)
    print(round(time.time() - t0, 1), sample(), flush=True)
print(proc.stdout.read()[-400:])
