import numpy as np
f32=np.float32
def seq(carry,d):
    out=np.empty(len(d),f32); acc=f32(carry)
    for i,x in enumerate(d):
        acc=f32(acc+x); out[i]=acc
    return out
def fast(carry,d):
    cb=np.array([carry],f32).view(np.uint32)[0]
    e=(cb>>23)&0xff
    if e<30 or e>=254: return None
    bb=cb&np.uint32(0xFF800000)
    B=np.array([bb],np.uint32).view(f32)[0]
    t=(B+d).astype(f32); ar=(t-B).astype(f32); err=(d-ar).astype(f32)
    half=np.array([bb-(24<<23)],np.uint32).view(f32)[0]
    ok=(d>=0)&(np.abs(err)!=half)&(t<f32(2)*B)
    if not ok.all(): return None
    a=np.minimum(t.view(np.uint32)-bb,1<<24).astype(np.int64)
    off=int(cb-bb)+np.cumsum(a)
    if off[-1]>=(1<<23): return None
    return (np.uint32(bb)+off.astype(np.uint32)).view(f32)
rng=np.random.default_rng(1)
tot=0;fb=0
for trial in range(3000):
    mode=trial%4
    n=64
    if mode==0: d=(rng.random(n)*7.5+0.5).astype(f32)/f32(4)
    elif mode==1: d=(rng.integers(0,64,n)/8).astype(f32)
    elif mode==2: d=(rng.random(n)*3).astype(f32); d[rng.random(n)<.1]=0
    else: d=(rng.integers(0,1<<12,n)).astype(f32)*f32(2.0**-rng.integers(0,14))
    carry=f32(rng.random()*2.0**rng.integers(-2,20))
    s=seq(carry,d); f=fast(carry,d); tot+=1
    if f is None: fb+=1; continue
    assert np.array_equal(s,f),(trial,carry,d[:5],s[:5],f[:5])
print("ok",tot,"fallbacks",fb)
def reasons(carry,d):
    cb=np.array([carry],f32).view(np.uint32)[0]
    e=(cb>>23)&0xff
    if e<30: return "start"
    bb=cb&np.uint32(0xFF800000)
    B=np.array([bb],np.uint32).view(f32)[0]
    t=(B+d).astype(f32); ar=(t-B).astype(f32); err=(d-ar).astype(f32)
    half=np.array([bb-(24<<23)],np.uint32).view(f32)[0]
    ties=(np.abs(err)==half).sum()
    if ties: return "tie%d"%min(ties,9)
    a=np.minimum(t.view(np.uint32)-bb,1<<24).astype(np.int64)
    off=int(cb-bb)+np.cumsum(a)
    if off[-1]>=(1<<23): return "cross"
    return "fast"
from collections import Counter
for N,keepfrac in ((100000,0.0),(100000,0.02),(10000,0.02),(1000,0.02)):
    x=(rng.integers(0,1<<24,N)).astype(f32)*f32(1/16777216)
    w=(f32(0.5)+f32(7.5)*x).astype(f32)
    P=np.empty(N,f32); acc=f32(0)
    for i in range(N): acc=f32(acc+w[i]); P[i]=acc
    d=np.diff(np.concatenate([[f32(0)],P])).astype(f32)
    keep=rng.random(N)<keepfrac
    wq=np.where(keep,d,(d/f32(4)).astype(f32)).astype(f32)
    c=Counter(); carry=f32(0)
    for j in range(0,N,64):
        ch=wq[j:j+64]
        c[reasons(carry,ch)]+=1
        carry=seq(carry,ch)[-1]
    print(N,keepfrac,dict(c))
