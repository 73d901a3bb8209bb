#!/usr/bin/env python3
"""Lane-level model of the consumer wavefront of k_challenge_sha256 (go-eth-kzg_amd/csrc/verify_kernels.hip).

One SHA-256 chain is run by a PAIR of lanes (mirror images within a row of 16): the "E" lane holds (e, f, g, h), the "A"
lane (a, b, c, d) and runs TWO rounds behind.  Every method of `Pair` below is one instruction of the kernel (same operands,
same lane masks: SHA_INIT_E, SHA_STEP, SHA_CORR0, SHA_BOUNDARY_A, SHA_END_BODY); running it against hashlib checks the
skewed schedule and its feed-forward corrections without a GPU (tests/test_sha_pair_model.py).  Reference for what is
hashed: internal/kzg/fiatshamir.go:22-33 (compute_challenge)."""
import hashlib, struct, os, random
M=0xffffffff
K=[0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2]
IV=[0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19]
def rotr(x,n): return ((x>>n)|(x<<(32-n)))&M
def schedule(block):
    w=list(struct.unpack('>16I',block))
    for i in range(16,64):
        s0=rotr(w[i-15],7)^rotr(w[i-15],18)^(w[i-15]>>3)
        s1=rotr(w[i-2],17)^rotr(w[i-2],19)^(w[i-2]>>10)
        w.append((w[i-16]+s0+w[i-7]+s1)&M)
    return [(w[i]+K[i])&M for i in range(64)]
# two lanes: 'E','A'; registers per lane
class Pair:
    def __init__(s):
        s.r={'E':{}, 'A':{}}
    def setl(s,lane,name,v): s.r[lane][name]=v&M
    def both(s,name,ve,va): s.r['E'][name]=ve&M; s.r['A'][name]=va&M
    # VALU ops (both lanes)
    def alignbit(s,d,a,sh):
        for l in 'EA': s.r[l][d]=rotr(s.r[l][a], s.r[l][sh])
    def x78(s,d,a,b,c):
        for l in 'EA': s.r[l][d]=s.r[l][a]^(s.r[l][b]&s.r[l][c])
    def xor3(s,d,a,b,c):
        for l in 'EA': s.r[l][d]=s.r[l][a]^s.r[l][b]^s.r[l][c]
    def ch(s,d,a,b,c):
        for l in 'EA': s.r[l][d]=(s.r[l][a]&s.r[l][b])|((~s.r[l][a])&s.r[l][c]&M)
    def add3(s,d,a,b,c):
        for l in 'EA': s.r[l][d]=(s.r[l][a]+s.r[l][b]+s.r[l][c])&M
    def xad(s,d,a,b,c):
        for l in 'EA': s.r[l][d]=((s.r[l][a]^s.r[l][b])+s.r[l][c])&M
    # masked DPP ops. src lane: 'self' (identity), 'fromA' (row_shl:4: E lanes read A lane), 'fromE' (row_shr:4: A lanes read E lane); bank: lanes written
    def _src(s,lane,mode,name):
        if mode=='id': return s.r[lane][name]
        if mode=='shl': return s.r['A'][name] if lane=='E' else None   # E reads its A partner
        if mode=='shr': return s.r['E'][name] if lane=='A' else None
        if mode=='mir': return s.r['A' if lane=='E' else 'E'][name]
    def add_dpp(s,d,src0,src1,mode,bank):
        new={}
        for l in bank:
            v=s._src(l,mode,src0); assert v is not None
            new[l]=(v+s.r[l][src1])&M
        for l in new: s.r[l][d]=new[l]
    def sub_dpp(s,d,src0,src1,mode,bank):
        new={}
        for l in bank:
            v=s._src(l,mode,src0); new[l]=(v-s.r[l][src1])&M
        for l in new: s.r[l][d]=new[l]
    def subrev_dpp(s,d,src0,src1,mode,bank):
        new={}
        for l in bank:
            v=s._src(l,mode,src0); new[l]=(s.r[l][src1]-v)&M
        for l in new: s.r[l][d]=new[l]
    def neg(s,d,a):
        for l in 'EA': s.r[l][d]=(-s.r[l][a])&M
    def mov_dpp(s,d,src0,mode,bank):
        new={}
        for l in bank: new[l]=s._src(l,mode,src0)
        for l in new: s.r[l][d]=new[l]


def step(P,R0,R1,R2,T,HWC,HWN,WKN):
    P.alignbit('ta',R0,'s1'); P.alignbit('tb',R0,'s2'); P.alignbit('tc',R0,'s3')
    P.x78('tx',R0,R2,'ma'); P.xor3('ta','ta','tb','tc'); P.ch('tx','tx',R1,R2)
    P.xad(HWN,R2,'ma',WKN)
    P.add3(T,'tx',HWC,'ta')
    P.add_dpp(HWN,R0,HWN,'mir','EA')
def corr0(P):   # after step 0 (next addend in hY)
    P.add_dpp('hY','hs2','hY','mir','E')
    P.add_dpp('hY','ho','hY','mir','A')
def boundary_A(P):
    # after step 1: state in (p2,p3,p0,p1); next addend in hX
    P.add_dpp('hX','hs1','hX','mir','E')
    P.sub_dpp('hX','hX','hs3','id','A')
    P.add_dpp('hs0','p2','hs0','id','A'); P.add_dpp('hs1','p3','hs1','id','A'); P.add_dpp('hs2','p0','hs2','id','A'); P.add_dpp('hs3','p1','hs3','id','A')
    P.mov_dpp('p2','hs0','id','A'); P.mov_dpp('p3','hs1','id','A'); P.mov_dpp('p0','hs2','id','A'); P.mov_dpp('p1','hs3','id','A')
names=['p0','p1','p2','p3']
def do_step(P,i):
    ph=i%4
    R0=names[(0-ph)%4]; R1=names[(1-ph)%4]; R2=names[(2-ph)%4]; T=names[(3-ph)%4]
    HWC,HWN=('hX','hY') if i%2==0 else ('hY','hX')
    step(P,R0,R1,R2,T,HWC,HWN,'w%d'%((i+1)&63))
def body(P,wk):
    for i in range(64): P.both('w%d'%i, wk[i], 1)
    P.mov_dpp('hX','p3','id','E'); P.add_dpp('hX','w0','hX','id','E'); P.add_dpp('hX','p1','hX','mir','E'); P.add_dpp('hX','hs3','hX','mir','E')
    for i in range(64):
        do_step(P,i)
        if i==0: corr0(P)
        if i==1: boundary_A(P)
    # end of body: E feed-forward, old hs0 kept for the A lane's correction
    P.neg('ho','hs0')
    P.add_dpp('hs0','p0','hs0','id','E'); P.add_dpp('hs1','p1','hs1','id','E'); P.add_dpp('hs2','p2','hs2','id','E'); P.add_dpp('hs3','p3','hs3','id','E')
    P.mov_dpp('p0','hs0','id','E'); P.mov_dpp('p1','hs1','id','E'); P.mov_dpp('p2','hs2','id','E'); P.mov_dpp('p3','hs3','id','E')
def epilogue(P):
    for i in range(64): P.both('w%d'%i, 0, 1)
    do_step(P,0); corr0(P); do_step(P,1); boundary_A(P)
def sha_pair(msg):
    ml=len(msg); msg=msg+b'\x80'+b'\0'*((55-ml)%64)+struct.pack('>Q',ml*8)
    P=Pair()
    P.both('ma',0,M); P.both('s1',6,2); P.both('s2',11,13); P.both('s3',25,22)
    P.both('p0',IV[4],0); P.both('p1',IV[5],0); P.both('p2',IV[6],0); P.both('p3',IV[7],0)
    P.both('hs0',IV[4],IV[0]); P.both('hs1',IV[5],IV[1]); P.both('hs2',IV[6],IV[2]); P.both('hs3',IV[7],IV[3])
    P.both('hX',0,0); P.both('hY',0,0); P.both('ho',-IV[4],0)
    for o in range(0,len(msg),64): body(P,schedule(msg[o:o+64]))
    epilogue(P)
    H=[P.r['A']['hs%d'%i] for i in range(4)]+[P.r['E']['hs%d'%i] for i in range(4)]
    return struct.pack('>8I',*H)
def selftest():
    for n in (0, 3, 55, 56, 64, 119, 120, 200, 1000):
        m = os.urandom(n)
        assert sha_pair(m) == hashlib.sha256(m).digest(), n
if __name__ == "__main__":
    selftest()
    print("model ok")
