"""Numpy twin of the dropout RNG (common.hpp: hash32 / hash32w) that ranked the candidates for the second-level hash: rate z-score,
largest row / column deviation, and neighbour correlations (in sigma) of a 4096 x 4096 keep mask.  "a" (one multiply round) fails by
hundreds of sigma, "d" (multiply, fold, multiply) is what hash32w is.  usage: python tools/rng_stats.py"""
import numpy as np, math
M32=np.uint64(0xffffffff)
def u32(x): return x & M32
def hash32(x):
    x=u32(x); x^=x>>np.uint64(16); x=u32(x*np.uint64(0x7feb352d)); x^=x>>np.uint64(15); x=u32(x*np.uint64(0x846ca68b)); x^=x>>np.uint64(16); return x
def hw_a(x):  # current hash32w
    x=u32(x); x^=x>>np.uint64(16); x=u32(x*np.uint64(0x7feb352d)); x^=x>>np.uint64(15); return x
def hw_b(x):  # two multiplies, no last xorshift
    x=u32(x); x^=x>>np.uint64(16); x=u32(x*np.uint64(0x7feb352d)); x^=x>>np.uint64(15); x=u32(x*np.uint64(0x846ca68b)); return x
def hw_c(x):  # mul, xs, mul  (no first xorshift), final xs16
    x=u32(x); x=u32(x*np.uint64(0x7feb352d)); x^=x>>np.uint64(15); x=u32(x*np.uint64(0x846ca68b)); x^=x>>np.uint64(16); return x
def hw_d(x):  # mul xs15 mul
    x=u32(x); x=u32(x*np.uint64(0x7feb352d)); x^=x>>np.uint64(15); x=u32(x*np.uint64(0x846ca68b)); return x
def run(hw, key, p, M=4096, N=4096):
    rows=np.arange(M,dtype=np.uint64); rk=hash32(rows ^ np.uint64(key))
    idx=np.arange(N//2,dtype=np.uint64)
    h=hw(rk[:,None]+idx[None,:])
    thr=int(p*65536)
    keep=np.empty((M,N),dtype=np.float32)
    keep[:,0::2]=((h&np.uint64(0xffff))>=thr); keep[:,1::2]=((h>>np.uint64(16))>=thr)
    q=1-thr/65536; sig=math.sqrt(q*(1-q)); n=M*N
    z=(keep.mean()-q)/(sig/math.sqrt(n))
    zr=np.abs(keep.mean(1)-q).max()/(sig/math.sqrt(N)); zc=np.abs(keep.mean(0)-q).max()/(sig/math.sqrt(M))
    c=keep-q
    corr=lambda a,b:(a*b).mean()/sig/sig*math.sqrt(n)
    return z,zr,zc,corr(c[:,:-1],c[:,1:]),corr(c[:,0::2],c[:,1::2])/math.sqrt(2),corr(c[:-1],c[1:]),corr(c[:,:-2],c[:,2:])
for name,hw in (("hash32",hash32),("a",hw_a),("b",hw_b),("c",hw_c),("d",hw_d)):
    for key in (0x1234567,0x9e3779b9,0xdeadbeef):
        for p in (0.1,0.25):
            print(name,hex(key),p," ".join(f"{v:6.2f}" for v in run(hw,key,p)))
print("---- d, more keys / p")
import random
random.seed(1)
worst=0
for t in range(8):
    key=random.getrandbits(32)
    for p in (0.05,0.1,0.2,0.5):
        r=run(hw_d,key,p)
        worst=max(worst,abs(r[0]),abs(r[3]),abs(r[4]),abs(r[5]),abs(r[6]))
        print(hex(key),p," ".join(f"{v:6.2f}" for v in r))
print("worst |z| (excluding row/col maxima)",worst)
