import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_dfcnn_transformer_amd import ops
N, T, C, H = 2, 512, 128, 2
g = torch.Generator(device='cuda').manual_seed(0)
Q, K, V, dO = [torch.randn(N, T, C, device='cuda', generator=g).relu_() for _ in range(4)]
V[:, 256:] = float('nan')
O = torch.zeros_like(Q); lse = torch.zeros(2, N, H, T, device='cuda')
ops.attention_fwd(Q, K, V, N, T, T, C, H, True, O, lse)
torch.cuda.synchronize()
for b in range(4):
    print('q block', b, 'finite:', bool(torch.isfinite(O[:, b*128:(b+1)*128]).all()))
