import numpy as np, torch, sys
sys.path.insert(0, '.')
from em_pose_amd import _lib
lib = _lib.lib()
def run(M,N,K,a,w):
    A=torch.tensor(a,dtype=torch.float32,device='cuda'); W=torch.tensor(w,dtype=torch.float32,device='cuda')
    out=torch.zeros(M,N,device='cuda')
    _lib.check(lib.empose_linear_f32(_lib.dptr(A),K,_lib.dptr(W),K,_lib.dptr(out),N,M,N,K,None,None,0,0.0,None))
    torch.cuda.synchronize()
    return out.cpu().numpy()
M,N,K=32,32,8
for k in range(8):
    a=np.zeros((M,K)); a[:,k]=1.0
    w=np.zeros((N,K)); w[:,:]=np.arange(K)[None,:]+1
    print('k',k, run(M,N,K,a,w)[0,:4])
a=np.arange(M*K).reshape(M,K)*1.0; w=np.eye(N,K)
print(run(M,N,K,a,w)[:3,:8]); print(a[:3])
