"""Error of an f32 dot product evaluated three ways against f64: f32 accumulation in k16 blocks (the MFMA model), bf16x3 pieces
(six products per MAC, conv_split3.hip) and fp16x2 pieces (three products per MAC of operands scaled by a power of two from the
tensor maximum, conv_h2.hip) -- the numbers behind DESIGN.md section 3 "fp16x2"; second part: rows far below the tensor maximum,
with and without f16 subnormals (gfx950 keeps them: mfma_f16_denorm_probe.hip).  numpy only:  python tools/experiments/fp16x2_error_sim.py"""
import numpy as np
rng=np.random.default_rng(0)
def bf16_trunc(x):
    u=x.view(np.uint32)&np.uint32(0xffff0000); return u.view(np.float32)
def split_bf3(x):
    h=bf16_trunc(x); r=(x-h).astype(np.float32); m=bf16_trunc(r); l=bf16_trunc((r-m).astype(np.float32)); return h,m,l
def split_h2(x,scale):
    xs=(x*np.float32(scale)).astype(np.float32)
    h=xs.astype(np.float16); r=(xs-h.astype(np.float32)).astype(np.float32); l=r.astype(np.float16)
    return h.astype(np.float32),l.astype(np.float32)
def blocked_dot(A,W,blk=16):
    # f32 accumulate in blocks of 16 (block sums in f64 then rounded: models MFMA internal exactness) 
    M,K=A.shape; acc=np.zeros((M,W.shape[1]),np.float32)
    for k in range(0,K,blk):
        acc=(acc+ (A[:,k:k+blk].astype(np.float64)@W[k:k+blk].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc
def pow2scale(mx): 
    e=np.floor(np.log2(mx)); return 2.0**(14-e)
for K,N,label in [(64,256,'K=64'),(256,256,'K=256'),(2304,256,'K=2304')]:
    M=512
    A=np.maximum(rng.standard_normal((M,K))*rng.lognormal(0,1.0,(M,1)),0).astype(np.float32)  # heavy tail across rows
    W=(rng.standard_normal((K,N))*np.sqrt(2/K)).astype(np.float32)
    ref=A.astype(np.float64)@W.astype(np.float64)
    f32=blocked_dot(A,W)
    h,m,l=split_bf3(A); wh,wm,wl=split_bf3(W)
    b3=np.zeros_like(f32)
    # products accumulated per k-block, smallest first like the kernel? just sum per block in f64 then f32 accumulate
    def acc_products(prods):
        acc=np.zeros((M,N),np.float32)
        for k in range(0,K,16):
            for (a,w) in prods:
                acc=(acc+(a[:,k:k+16].astype(np.float64)@w[k:k+16].astype(np.float64)).astype(np.float32)).astype(np.float32)
        return acc
    b3=acc_products([(l,wh),(m,wm),(h,wl),(m,wh),(h,wm),(h,wh)])
    sa=pow2scale(np.abs(A).max()); sw=pow2scale(np.abs(W).max())
    ah,al=split_h2(A,sa); whh,wll=split_h2(W,sw)
    h2=acc_products([(al,whh),(ah,wll),(ah,whh)])/np.float32(sa*sw)
    den=np.abs(A.astype(np.float64))@np.abs(W.astype(np.float64))
    for name,y in [('f32',f32),('bf16x3',b3),('fp16x2',h2)]:
        err=np.abs(y-ref)
        print(label,name,'max err/den %.3e  rms err/den %.3e'%((err/den).max(), np.sqrt(((err/den)**2).mean())))
print('--- dynamic range: rows scaled by 2^-j relative to the tensor max')
K,N,M=256,64,16*8
A=np.maximum(rng.standard_normal((M,K)),0).astype(np.float32)
for g in range(8): A[g*16:(g+1)*16]*=np.float32(2.0**(-5*g))
W=(rng.standard_normal((K,N))*np.sqrt(2/K)).astype(np.float32)
ref=A.astype(np.float64)@W.astype(np.float64); den=np.abs(A.astype(np.float64))@np.abs(W.astype(np.float64))
def accp(prods):
    acc=np.zeros((M,N),np.float32)
    for k in range(0,K,16):
        for (a,w) in prods:
            acc=(acc+(a[:,k:k+16].astype(np.float64)@w[k:k+16].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc
sa=pow2scale(np.abs(A).max()); sw=pow2scale(np.abs(W).max())
ah,al=split_h2(A,sa); whh,wll=split_h2(W,sw)
h2=accp([(al,whh),(ah,wll),(ah,whh)])/np.float32(sa*sw)
f32=blocked_dot(A,W)
# flush-to-zero variant of subnormal pieces
def ftz(x): 
    y=x.copy(); y[np.abs(y)<2.0**-14]=0; return y
h2f=accp([(ftz(al),ftz(whh)),(ftz(ah),ftz(wll)),(ftz(ah),ftz(whh))])/np.float32(sa*sw)
for g in range(8):
    s=slice(g*16,(g+1)*16)
    print('rows x 2^-%d: f32 %.2e  fp16x2 %.2e  fp16x2-ftz %.2e (max err/den)'%(5*g,(np.abs(f32-ref)/den)[s].max(),(np.abs(h2-ref)/den)[s].max(),(np.abs(h2f-ref)/den)[s].max()))
