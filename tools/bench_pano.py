import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
print(json.dumps(bench.pano_stretch_leg(torch.device("cuda:0"))))
