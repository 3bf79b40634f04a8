import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import nmfgpu_amd as na
print("device_count via lib:", na.device_count())
import torch
print("torch avail:", torch.cuda.is_available())
for l in open("/proc/self/maps"):
    if "amdhip" in l or "hsa-runtime" in l or "libhsakmt" in l:
        if "r-xp" in l: print(l.split()[-1])
