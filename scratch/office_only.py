import sys, json, torch
sys.path.insert(0, '.')
import bench
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
class A: pass
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
r = bench.other_configs(dev, A(), steps=steps, only=("resnet50_dann_8w8a_b28",))
print(json.dumps(r["resnet50_dann_8w8a_b28"]))
