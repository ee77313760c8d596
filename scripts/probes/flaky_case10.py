import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.chdir('/root/repo')
import test_gpu_kernels as t
case = t.CONV_CASES[10]
bad = 0
for i in range(40):
    try:
        t.conv_case(case, torch.bfloat16)
    except AssertionError as e:
        bad += 1
        print(i, str(e)[:300], flush=True)
print('failures', bad, 'of 40')
