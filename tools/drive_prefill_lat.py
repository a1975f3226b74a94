import torch, sys
sys.path.insert(0, '.')
import bench
from torch import nn
from palu_amd.kernel.palu_attention import LatentCache, LlamaPaluAttention, build_b
H,G,D,GS,HIDDEN=32,8,128,4,4096
class Cfg: pass
cfg=Cfg(); cfg.hidden_size, cfg.num_attention_heads, cfg.attention_bias = HIDDEN, H, False
cfg.group_size, cfg.num_groups, cfg.total_rank_k, cfg.total_rank_v = GS, G, 1024, 3072
torch.manual_seed(0)
dev=torch.device('cuda:0')
with torch.device(dev):
    m = LlamaPaluAttention(cfg, 0).half()
    with torch.no_grad():
        for lin in (m.q_proj, m.k_proj.VT, m.v_proj.VT, m.o_proj): lin.weight.normal_(0.0, 0.02)
        for u in m.k_proj.U_list: u.weight.normal_(0.0, 128 ** -0.5)
    m.k_proj.B = nn.Parameter(build_b([u.weight for u in m.k_proj.U_list], GS, D))
m = m.eval().prepare_decode()
T=65536
x = torch.randn(1, T, HIDDEN, device=dev, dtype=torch.float16)
for _ in range(2):
    c = LatentCache(capacity=T+512)
    c.reserve(0, T+512, torch.empty((1,G,0,128),dtype=torch.float16,device=dev), torch.empty((1,G,0,384),dtype=torch.float16,device=dev))
    with torch.no_grad():
        out,_,_ = m(x, past_key_value=c, is_causal=True)
    torch.cuda.synchronize()
