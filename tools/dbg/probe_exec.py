"""Can a GPU-initialised python start a child process (fork+exec) on the GPU pool?"""
import subprocess, sys, torch
print('avail', torch.cuda.is_available())
x = torch.zeros(4, device='cuda'); torch.cuda.synchronize()
try:
    r = subprocess.run([sys.executable, '-c', 'print("child ok")'], capture_output=True, text=True, timeout=120)
    print('rc', r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
except Exception as e:
    print('EXC', repr(e))
