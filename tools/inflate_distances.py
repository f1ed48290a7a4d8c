"""Distance histogram of the DEFLATE matches in the synthetic observation BCF (ring size choice of the inflate kernel)."""
import sys, struct, zlib, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from varlociraptor_amd import synth, ingest
cfg = synth.config3()
b = synth.generate(cfg, 300, seed=1)
path='/tmp/obs_hist.bcf'
ingest.write_observations(path, b, 1)
raw=open(path,'rb').read()
# minimal inflate with match recording
class BR:
    def __init__(s,d): s.d=d; s.p=0; s.b=0; s.n=0
    def need(s,k):
        while s.n<k:
            s.b|=s.d[s.p]<<s.n; s.p+=1; s.n+=8
    def get(s,k):
        if k==0: return 0
        s.need(k); v=s.b&((1<<k)-1); s.b>>=k; s.n-=k; return v
def mk(lens):
    # canonical decode table: dict (len,code)->sym
    cnt=[0]*16
    for l in lens: cnt[l]+=1
    cnt[0]=0; code=0; nxt=[0]*16
    for i in range(1,16): code=(code+cnt[i-1])<<1; nxt[i]=code
    t={}
    for s,l in enumerate(lens):
        if l: t[(l,nxt[l])]=s; nxt[l]+=1
    return t
def dec(br,t):
    code=0
    for l in range(1,16):
        code=(code<<1)|br.get(1)
        if (l,code) in t: return t[(l,code)]
    raise ValueError
LB=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LX=[0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DB=[1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DX=[0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
dists=[];lens_=[];nlit=0
off=0;nm=0
while off<len(raw) and nm<40:
    bsize=struct.unpack_from('<H',raw,off+16)[0]+1
    data=raw[off+18:off+bsize-8]; off+=bsize; nm+=1
    if nm<3: continue
    br=BR(data+b'\0'*8); out=0
    while True:
        last=br.get(1); typ=br.get(2)
        if typ==0:
            br.b=0;br.n=0; ln=struct.unpack_from('<H',data,br.p)[0]; br.p+=4+ln; out+=ln
        else:
            if typ==1:
                ll=[8]*144+[9]*112+[7]*24+[8]*8; dl=[5]*30
            else:
                hlit=br.get(5)+257; hd=br.get(5)+1; hc=br.get(4)+4
                order=[16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15]; cl=[0]*19
                for i in range(hc): cl[order[i]]=br.get(3)
                ct=mk(cl); L=[]
                while len(L)<hlit+hd:
                    s=dec(br,ct)
                    if s<16: L.append(s)
                    elif s==16: L+= [L[-1]]*(3+br.get(2))
                    elif s==17: L+=[0]*(3+br.get(3))
                    else: L+=[0]*(11+br.get(7))
                ll=L[:hlit]; dl=L[hlit:]
            lt=mk(ll); dt=mk(dl)
            while True:
                s=dec(br,lt)
                if s<256: nlit+=1; out+=1
                elif s==256: break
                else:
                    k=s-257; ln=LB[k]+br.get(LX[k]); d=dec(br,dt); di=DB[d]+br.get(DX[d])
                    dists.append(di); lens_.append(ln); out+=ln
        if last: break
d=np.array(dists); l=np.array(lens_)
print("members",nm-2,"matches",len(d),"literals",nlit,"mean len %.1f"%l.mean())
for th in (1024,2048,4096,8192-330,16384-330,32768):
    print("dist > %5d: %.2f%% of matches, %.2f%% of matched bytes"%(th,100*(d>th).mean(),100*l[d>th].sum()/l.sum()))
