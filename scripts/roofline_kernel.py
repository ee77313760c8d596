"""Launch only the roofline kernel of bench.py (conv_igemm bf16, 3x3 64->64 @80x80, batch 64) a few times: the target of
the rocprofv3 --pmc passes whose FETCH_SIZE / WRITE_SIZE feed `roofline.traffic`."""
import os, sys
os.environ.setdefault('YOLOv5_VERBOSE', 'false')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(bench.conv_roofline(torch.device('cuda', 0), iters=10))
