#!/usr/bin/env python3
"""Does a process that has initialised the GPU get to start child processes on this box?  (tests that spawn ranks)"""
import subprocess
import sys

import torch

torch.cuda.init()
x = torch.ones(4, device='cuda').sum().item()
r = subprocess.run([sys.executable, '-c', 'import torch; print("child sees", torch.cuda.device_count(), "gpu(s);", '
                    'torch.ones(3, device="cuda").sum().item())'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print('parent ok', x, '| child rc', r.returncode, '|', r.stdout.strip()[-300:])
