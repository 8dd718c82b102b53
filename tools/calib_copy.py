"""Known-traffic kernel for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950:
three device-to-device copies of a 1 GiB fp32 tensor (reads 1 GiB, writes 1 GiB each)."""
import torch
x = torch.randn(256 * 1024 * 1024, device="cuda")
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
print("copied", x.numel() * 4, "bytes x3")
