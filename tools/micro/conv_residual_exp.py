import sys, torch, torch.nn as nn
sys.path.insert(0, '/root/repo')
from loftr_amd import ops
dev = torch.device('cuda', 0)
torch.manual_seed(0)
B, h, w, c = 16, 240, 320, 128
conv = nn.Conv2d(c, c, 3, 1, 1, bias=False).to(dev); bn = nn.BatchNorm2d(c).to(dev).eval()
x = ops.sp_from_nhwc(torch.relu(torch.randn(B, h, w, c, device=dev)))
r = ops.sp_from_nhwc(torch.relu(torch.randn(B, h, w, c, device=dev)))
rz = ops.sp_from_nhwc(torch.zeros(B, h, w, c, device=dev))
small = ops.sp_from_nhwc(torch.relu(torch.randn(1, h, w, c, device=dev)))
def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    print('no residual      %.1f us' % t(lambda: ops.conv_bn_act(x, c, conv, bn, act=1, residual=None, want_sp=True)))
    print('residual = other %.1f us' % t(lambda: ops.conv_bn_act(x, c, conv, bn, act=1, residual=r, want_sp=True)))
    print('residual = x     %.1f us' % t(lambda: ops.conv_bn_act(x, c, conv, bn, act=1, residual=x, want_sp=True)))
    print('residual = zeros %.1f us' % t(lambda: ops.conv_bn_act(x, c, conv, bn, act=1, residual=rz, want_sp=True)))
