"""Design-time simulation: dependency rounds per 64-sequence batch under different readiness rules."""
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import helpers as H, struct, numpy as np
from cases import text
d=text(2<<20)
s=H.oracle_compress(d,131072)
def parse_blocks(s):
    i=0
    while i<len(s):
        c=struct.unpack_from('<I',s,i+8)[0]; f=i+12; p=f+15
        while True:
            bh=struct.unpack_from('<I',s,p)[0]; p+=4
            if bh==0: break
            bs=bh&0x7fffffff
            if bh>>31: p+=bs; continue
            e=p+bs; seqs=[]
            while p<e:
                t=s[p]; p+=1; l=t>>4
                if l==15:
                    while True:
                        b=s[p]; p+=1; l+=b
                        if b!=255: break
                p+=l
                if p>=e: seqs.append((l,0,0)); break
                o=s[p]|s[p+1]<<8; p+=2; m=t&15
                if m==15:
                    while True:
                        b=s[p]; p+=1; m+=b
                        if b!=255: break
                seqs.append((l,m+4,o))
            yield seqs
        i+=12+c
R=8192
rounds_exact=[];rounds_wm=[];far=0;near=0;inb=0;tot=0
maxlit=[];maxml=[];span=[]
lit8=[];ml8=[]
for seqs in parse_blocks(s):
    for b0 in range(0,len(seqs),64):
        bt=seqs[b0:b0+64]
        n=len(bt)
        op=np.zeros(n+1,int)
        for k,(l,m,o) in enumerate(bt): op[k+1]=op[k]+l+m
        span.append(op[n]); maxlit.append(max(x[0] for x in bt)); maxml.append(max(x[1] for x in bt))
        lit8.append(max((x[0]+7)//8 for x in bt)); ml8.append(max((x[1]+7)//8 for x in bt))
        # dependencies
        deps=[]
        for k,(l,m,o) in enumerate(bt):
            if m==0: deps.append(None); continue
            mpos=op[k]+l; src=mpos-o; tot+=1
            ln=min(m,o)
            if src>=0:
                inb+=1
                a=np.searchsorted(op,src,side='right')-1; b=np.searchsorted(op,src+ln-1,side='right')-1
                deps.append((a,min(b,k-1),src+ln))
            else:
                deps.append((-1,-1,0))
                if o>R-2048: far+=1
                else: near+=1
        # exact rounds
        fin=[dd is None for dd in deps]; r=0
        fin=np.array(fin)
        while not fin.all():
            r+=1; newf=fin.copy()
            for k,dd in enumerate(deps):
                if fin[k]: continue
                a,b,_=dd
                if a<0 or b<a or fin[a:b+1].all(): newf[k]=True
            fin=newf
        rounds_exact.append(r)
        fin=np.array([dd is None for dd in deps]); r=0
        while not fin.all():
            r+=1; newf=fin.copy()
            w=op[np.argmin(fin)] if not fin.all() else op[n]
            first=np.argmin(fin)
            for k,dd in enumerate(deps):
                if fin[k]: continue
                a,b,e=dd
                if a<0 or e<=w or k==first: newf[k]=True
            fin=newf
        rounds_wm.append(r)
print("batches",len(span),"span mean %.0f max %d"%(np.mean(span),max(span)))
print("maxlit per batch mean %.1f p90 %d max %d ; maxml mean %.1f p90 %d max %d"%(np.mean(maxlit),np.percentile(maxlit,90),max(maxlit),np.mean(maxml),np.percentile(maxml,90),max(maxml)))
print("8B-iterations lit mean %.2f ml mean %.2f"%(np.mean(lit8),np.mean(ml8)))
print("matches: in-batch %.1f%% near(ring) %.1f%% far(global) %.1f%%"%(100*inb/tot,100*near/tot,100*far/tot))
print("rounds exact: mean %.2f p90 %d max %d"%(np.mean(rounds_exact),np.percentile(rounds_exact,90),max(rounds_exact)))
print("rounds watermark: mean %.2f p90 %d max %d"%(np.mean(rounds_wm),np.percentile(rounds_wm,90),max(rounds_wm)))
