import torch, sys
sys.path.insert(0, ".")
from mad_amd import ops
dev = torch.device("cuda:0")
for (B, R, Ci, Co) in [(32, 64, 256, 256), (32, 128, 128, 128), (32, 32, 512, 512), (32, 16, 1024, 512)]:
    x = torch.randn(B, R, R, Ci, device=dev).bfloat16()
    dy = torch.randn(B, R, R, Co, device=dev).bfloat16()
    dw = torch.zeros(Co, Ci, 3, 3, device=dev)
    for _ in range(3): ops.conv_wgrad(x, dy, 3, out=dw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.conv_wgrad(x, dy, 3, out=dw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(B, R, Ci, Co, "%.1f us %.0f TF" % (ms * 1e3, 2 * B * R * R * Ci * Co * 9 / ms / 1e9))
