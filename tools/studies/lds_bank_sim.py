import numpy as np, itertools
half=64; m2=128; n1=127
def tri(i): return i*(i+1)//2
def sym(i,j): return tri(i)+j if i>=j else tri(j)+i
def shift(u): return n1 if u==n1 else (n1-1 if u==0 else u-1)
# designated
des=set()
for k in range(half):
    if k==0: des.add((1,0))
    elif k==1: des.add((2,0))
    elif k<=half-2: des.add((k+1,k-1))
    else: des.add((half-1,half-2))
gen=[(al,be) for al in range(1,half) for be in range(al) if (al,be) not in des]
ngen=len(gen); per=960; hfront=(ngen+1)//2
def addrs(al,be,addrfn):
    pa,qa,pb,qb=al,n1-al,be,n1-be
    rd=[addrfn(pa,pb),addrfn(pa,qb),addrfn(qa,pb),addrfn(qa,qb)]
    pa2,qa2,pb2,qb2=map(shift,(pa,qa,pb,qb))
    wr=[addrfn(pa2,pb2),addrfn(pa2,qb2),addrfn(qa2,pb2),addrfn(qa2,qb2)]
    return rd,wr
def cost(lane_addrs, group, mod):
    # lane_addrs: list of 64 (addr or None)
    cyc=0
    for g0 in range(0,64,group):
        banks={}
        for a in lane_addrs[g0:g0+group]:
            if a is None: continue
            banks.setdefault(a%mod,set()).add(a)
        cyc+= max([len(v) for v in banks.values()],default=0) if banks else 0
        if not banks: cyc+=0
    return cyc
def evaluate(assign, addrfn, label):
    # assign: dict (wave,slot)->list of 64 blocks or None
    tot_r=tot_w=ideal_r=ideal_w=0
    for key,blocks in assign.items():
        A=[addrs(*b,addrfn) if b else None for b in blocks]
        for e in range(4):
            la=[a[0][e] if a else None for a in A]
            tot_r+=cost(la,32,32); ideal_r+=sum(1 for g0 in (0,32) if any(x is not None for x in la[g0:g0+32]))
            lw=[a[1][e] if a else None for a in A]
            tot_w+=cost(lw,16,16); ideal_w+=sum(1 for g0 in range(0,64,16) if any(x is not None for x in lw[g0:g0+16]))
    print(f"{label}: read cycles {tot_r} (ideal {ideal_r}), write array cycles {tot_w} (ideal {ideal_w})")
def current():
    assign={}
    for wave in range(16):
        for slot in range(2):
            blocks=[]
            for lane in range(64):
                tid=wave*64+lane; t=tid-64; b=None
                if wave==0:
                    if slot==0:
                        k=lane
                        b=sorted(des)[0] if False else None
                        if k==0: b=(1,0)
                        elif k==1: b=(2,0)
                        elif k<=half-2: b=(k+1,k-1)
                        else: b=(half-1,half-2)
                    else:
                        g=per+lane
                        b=gen[g] if g<ngen-per else None
                else:
                    if slot==0: b=gen[t] if t<hfront else None
                    else:
                        g=ngen-1-t
                        b=gen[g] if g>=hfront else None
                blocks.append(b)
            assign[(wave,slot)]=blocks
    return assign
evaluate(current(), sym, "packed-tri, current map")
LD=65
def rfp(i,j):
    if i<j: i,j=j,i
    if j<64: return (i+1)*LD+j
    return (j-64)*LD+(i-64)
evaluate(current(), rfp, "RFP LD=65, current map")

# ---- folded-row enumeration: wave-slot s (0..31): lanes [0,s): row s (be = lane) ; lanes [s,63): row 63-s (be = lane - s); designated blocks removed -> idle lanes
def folded(exclude_des=True):
    assign={}
    ws=0
    for s in range(32):
        blocks=[]
        for lane in range(64):
            if lane < s: b=(s,lane)
            elif lane < 63: b=(63-s, lane-s)
            else: b=None
            if b and (b[0]<=b[1]): b=None
            if b and exclude_des and b in des: b=None
            blocks.append(b)
        assign[(s//2, s%2+10*0, s)]=blocks
    # designated slot
    blocks=[]
    for k in range(64):
        if k==0: b=(1,0)
        elif k==1: b=(2,0)
        elif k<=half-2: b=(k+1,k-1)
        else: b=(half-1,half-2)
        blocks.append(b)
    assign[('des',)]=blocks
    return assign
for name,fn in (("packed-tri",sym),("RFP65",rfp)):
    evaluate(folded(), fn, name+", folded rows")
# layout with separate quadrants and chosen LDs
def quad(LDR, LDB):
    BT0 = 65*LDR
    BT0 += (-BT0)%32
    def f(i,j):
        if i<j: i,j=j,i
        if i<64: return (i+1)*LDR + j                 # TT
        if j>=64: return (j-64)*LDR + (i-64)          # BB transposed into the same rectangle
        return BT0 + (i-64)*LDB + j                   # BT square
    return f
for LDR,LDB in ((65,65),(67,67),(65,67),(69,69),(65,69),(73,73),(65,73)):
    evaluate(folded(), quad(LDR,LDB), f"quad LDR={LDR} LDB={LDB}, folded")
    evaluate(current(), quad(LDR,LDB), f"quad LDR={LDR} LDB={LDB}, current")
