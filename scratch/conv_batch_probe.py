import torch, time
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda')
cl = torch.channels_last
shapes = [(64, 64, 56, 1, 1), (64, 64, 56, 3, 1), (64, 256, 56, 1, 1), (256, 64, 56, 1, 1), (128, 128, 28, 3, 1), (512, 128, 28, 1, 1), (256, 256, 14, 3, 1), (1024, 256, 14, 1, 1), (512, 512, 7, 3, 1), (2048, 512, 7, 1, 1)]
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
tot = {28: 0.0, 56: 0.0}
for cin, cout, hw, k, s in shapes:
    conv = torch.nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(dev).to(memory_format=cl)
    res = {}
    for B in (28, 56):
        x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=cl).requires_grad_(True)
        def step():
            y = conv(x)
            y.backward(torch.ones_like(y))
        res[B] = t(step)
    tot[28] += 2 * res[28]; tot[56] += res[56]
    print(f"cin {cin} cout {cout} hw {hw} k {k}: 2 x batch 28 = {2 * res[28]:.0f} us, batch 56 = {res[56]:.0f} us")
print("sum:", tot)
