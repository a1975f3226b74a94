import torch, math
from palu_amd import _lib
lib=_lib
def gemv(W,x):
    N,K=W.shape
    y=torch.full((N,),-7.0,dtype=torch.float16,device="cuda")
    lib.check(lib.lib.palu_gemv_f16(W.data_ptr(), W.stride(0), x.data_ptr(), y.data_ptr(), N, K, torch.cuda.current_stream().cuda_stream),"g")
    return y
K=512; N=2
W=(torch.arange(K,device="cuda").float()+1).reshape(1,K).expand(N,K).half().contiguous()
bad=[]
for j in range(K):
    x=torch.zeros(K,dtype=torch.float16,device="cuda"); x[j]=1
    y=gemv(W,x)
    if abs(float(y[0])-(j+1))>0.5: bad.append((j,float(y[0])))
print(len(bad), bad[:40])
