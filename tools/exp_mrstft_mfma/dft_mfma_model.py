"""numpy model of the matrix-core DFTs of csrc/mrstft.hip: the v_mfma_f32_32x32x16_f16 lane layouts (A: row = lane & 31,
k = 8 (lane >> 5) + j; B: column = lane & 31, same k; D: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column = lane & 31)
and the index algebra of the forward / inverse 512-, 1024- and 2048-point transforms in fp64, against numpy.fft.
    python tools/probe/dft_mfma_model.py"""
import numpy as np
# --- wave-level MFMA 32x32x16 emulation (layouts as used by csrc/conv_f16.hip) ---
def mfma(A, B, C):
    """A, B: (64, 8) per-lane fragments; C: (64, 16). D[m][n] += sum_k A[m][k] B[k][n];
    A lane l: row m = l&31, k = 8*(l>>5)+j ; B lane l: col n = l&31, k = 8*(l>>5)+j ;
    C lane l, reg r: row (r&3)+8*(r>>2)+4*(l>>5), col l&31."""
    Am = np.zeros((32,16), A.dtype); Bm = np.zeros((16,32), B.dtype)
    for l in range(64):
        for j in range(8):
            Am[l&31, 8*(l>>5)+j] = A[l,j]
            Bm[8*(l>>5)+j, l&31] = B[l,j]
    D = Am @ Bm
    out = C.copy()
    for l in range(64):
        for r in range(16):
            out[l,r] += D[(r&3)+8*(r>>2)+4*(l>>5), l&31]
    return out
def row_of(r, hh): return (r&3)+8*(r>>2)+4*hh
def slot_index(q, j, hh):        # the index held by k-slot (q, j) of lane-half hh  == row_of(8q+j, hh)
    return row_of(8*q+j, hh)
N=1024
W=lambda n,e: np.exp(-2j*np.pi*e/n)
lanes=np.arange(64); idx=lanes&31; HH=lanes>>5
# constant fragments Fq[q][lane][j] = W32^{ slot_index(q,j,hh) * idx }  (serves step 1 as B and step 2 as A)
F=[np.array([[W(32, slot_index(q,j,HH[l])*idx[l]) for j in range(8)] for l in range(64)]) for q in range(2)]
def cmfma(Ar,Ai,Br,Bi,Cr,Ci):
    Cr=mfma(Ar,Br,Cr); Cr=mfma(-Ai,Bi,Cr); Ci=mfma(Ar,Bi,Ci); Ci=mfma(Ai,Br,Ci); return Cr,Ci
def fwd1024(z):
    # step 1: A = data: lane (n2 = idx, hh): slot (q,j) holds z[32*n1 + n2], n1 = slot_index(q,j,hh)
    Tr=np.zeros((64,16)); Ti=np.zeros((64,16))
    for q in range(2):
        A=np.array([[z[32*slot_index(q,j,HH[l])+idx[l]] for j in range(8)] for l in range(64)])
        Tr,Ti=cmfma(A.real,A.imag,F[q].real,F[q].imag,Tr,Ti)
    # D layout: lane col = k1 = idx, rows n2 = row_of(r,hh); twiddle W1024^{n2*k1}
    T=Tr+1j*Ti
    for l in range(64):
        for r in range(16):
            T[l,r]*=W(1024, row_of(r,HH[l])*idx[l])
    # step 2: A = F (m = k2 = idx, slots n2), B = T' (slot (q,j) = reg r = 8q+j)
    Zr=np.zeros((64,16)); Zi=np.zeros((64,16))
    for q in range(2):
        B=T[:,8*q:8*q+8]
        Zr,Zi=cmfma(F[q].real,F[q].imag,B.real,B.imag,Zr,Zi)
    Z=Zr+1j*Zi
    # D layout: rows k2 = row_of(r,hh), col k1 = idx: bin k = k1 + 32*k2
    out=np.zeros(N,complex)
    for l in range(64):
        for r in range(16): out[idx[l]+32*row_of(r,HH[l])]=Z[l,r]
    return out, Z
rng=np.random.default_rng(0)
z=rng.standard_normal(N)+1j*rng.standard_normal(N)
out,Zregs=fwd1024(z)
print("fwd1024 err", np.abs(out-np.fft.fft(z)).max())
# inverse: g[n] = Re sum_{k<=N/2} G[k] e^{+2 pi i k n/N}; input regs = forward-output layout (lane k1=k&31, reg r <-> k2=row_of(r,hh))
def inv1024(Gfull):   # Gfull: length N complex, zero above N/2
    Fc=[f.conj() for f in F]
    G=np.zeros((64,16),complex)
    for l in range(64):
        for r in range(16): G[l,r]=Gfull[idx[l]+32*row_of(r,HH[l])]
    # step 1: A = G data (m = kb = idx, slot (q,j) = ka = slot_index = row_of(8q+j,hh) -> exactly reg r=8q+j), B = Fc
    Ur=np.zeros((64,16)); Ui=np.zeros((64,16))
    q=0
    A=G[:,0:8]
    Ur,Ui=cmfma(A.real,A.imag,Fc[q].real,Fc[q].imag,Ur,Ui)
    U=Ur+1j*Ui   # rows kb = row_of(r,hh), col na = idx
    # k = 512 term (ka = 16 = row_of(8,0), kb = 0): lane 0 reg 8 ; contributes G512 * conjW32^{16*na} = G512*(-1)^na at row kb=0
    g512=G[0,8]
    for l in range(64):
        for r in range(16):
            if row_of(r,HH[l])==0: U[l,r]+=g512*((-1.0)**idx[l])
    for l in range(64):
        for r in range(16): U[l,r]*=np.conj(W(1024,row_of(r,HH[l])*idx[l]))
    # step 2: g[nb][na] = Re sum_kb Fc[nb][kb] U'[kb][na]
    gr=np.zeros((64,16))
    for q in range(2):
        B=U[:,8*q:8*q+8]
        gr=mfma(Fc[q].real,B.real,gr); gr=mfma(-Fc[q].imag,B.imag,gr)
    out=np.zeros(N)
    for l in range(64):
        for r in range(16): out[idx[l]+32*row_of(r,HH[l])]=gr[l,r]
    return out
Gs=np.zeros(N,complex); Gs[:N//2+1]=rng.standard_normal(N//2+1)+1j*rng.standard_normal(N//2+1)
ref=np.real(np.fft.ifft(Gs)*N)
print("inv1024 err", np.abs(inv1024(Gs)-ref).max())

# ---------------- 512: two frames per wave, block-diagonal 16-point step ----------------
BD=[np.array([[ (W(16,(idx[l]&15)*(slot_index(q,j,HH[l])&15)) if (idx[l]>>4)==(slot_index(q,j,HH[l])>>4) else 0.0) for j in range(8)] for l in range(64)]) for q in range(2)]
def fwd512(z2):     # z2: (2, 512) complex (real in practice)
    Tr=np.zeros((64,16)); Ti=np.zeros((64,16))
    for q in range(2):
        A=np.array([[z2[idx[l]>>4, 16*slot_index(q,j,HH[l])+(idx[l]&15)] for j in range(8)] for l in range(64)])
        Tr,Ti=cmfma(A.real,A.imag,F[q].real,F[q].imag,Tr,Ti)
    T=Tr+1j*Ti      # lane col k1 = idx, rows (f,n2) = row_of(r,hh)
    for l in range(64):
        for r in range(16): T[l,r]*=W(512,(row_of(r,HH[l])&15)*idx[l])
    Zr=np.zeros((64,16)); Zi=np.zeros((64,16))
    for q in range(2):
        B=T[:,8*q:8*q+8]
        Zr,Zi=cmfma(BD[q].real,BD[q].imag,B.real,B.imag,Zr,Zi)
    Z=Zr+1j*Zi      # lane col k1, rows (f,k2): bin k = k1 + 32*k2
    out=np.zeros((2,512),complex)
    for l in range(64):
        for r in range(16):
            row=row_of(r,HH[l]); out[row>>4, idx[l]+32*(row&15)]=Z[l,r]
    return out,Z
z2=rng.standard_normal((2,512))+1j*rng.standard_normal((2,512))
o512,_=fwd512(z2)
print("fwd512 err", np.abs(o512-np.fft.fft(z2,axis=1)).max())
def inv512(G2):     # G2 (2,512) complex, zero above 256; returns (2,512) real = Re sum_k G e^{+i...}
    G=np.zeros((64,16),complex)
    for l in range(64):
        for r in range(16):
            row=row_of(r,HH[l]); G[l,r]=G2[row>>4, idx[l]+32*(row&15)]
    # step 1: A = G (lane row m = k1, slots (f,k2) = regs), B = BD16 (col n = (f,nb)); U = G * conj(BD)
    Ur=np.zeros((64,16)); Ui=np.zeros((64,16))
    for q in range(2):
        A=G[:,8*q:8*q+8]; C=BD[q]
        Ur=mfma(A.real,C.real,Ur); Ur=mfma(A.imag,C.imag,Ur); Ui=mfma(A.imag,C.real,Ui); Ui=mfma(-A.real,C.imag,Ui)
    U=Ur+1j*Ui      # lane col (f,nb), rows k1
    for l in range(64):
        for r in range(16): U[l,r]*=np.conj(W(512,row_of(r,HH[l])*(idx[l]&15)))
    g=np.zeros((64,16))
    for q in range(2):
        B=U[:,8*q:8*q+8]
        g=mfma(F[q].real,B.real,g); g=mfma(F[q].imag,B.imag,g)
    out=np.zeros((2,512))
    for l in range(64):
        for r in range(16): out[idx[l]>>4, 16*row_of(r,HH[l])+(idx[l]&15)]=g[l,r]
    return out
G2=np.zeros((2,512),complex); G2[:,:257]=rng.standard_normal((2,257))+1j*rng.standard_normal((2,257))
print("inv512 err", np.abs(inv512(G2)-np.real(np.fft.ifft(G2,axis=1)*512)).max())
# also 1024 inverse written in the generic (no K skip) form: U = G conj(F) with all slots
def inv1024_generic(Gfull):
    G=np.zeros((64,16),complex)
    for l in range(64):
        for r in range(16): G[l,r]=Gfull[idx[l]+32*row_of(r,HH[l])]
    Ur=np.zeros((64,16)); Ui=np.zeros((64,16))
    for q in range(2):
        A=G[:,8*q:8*q+8]; C=F[q]
        Ur=mfma(A.real,C.real,Ur); Ur=mfma(A.imag,C.imag,Ur); Ui=mfma(A.imag,C.real,Ui); Ui=mfma(-A.real,C.imag,Ui)
    U=Ur+1j*Ui
    for l in range(64):
        for r in range(16): U[l,r]*=np.conj(W(1024,row_of(r,HH[l])*idx[l]))
    g=np.zeros((64,16))
    for q in range(2):
        B=U[:,8*q:8*q+8]
        g=mfma(F[q].real,B.real,g); g=mfma(F[q].imag,B.imag,g)
    out=np.zeros(N)
    for l in range(64):
        for r in range(16): out[idx[l]+32*row_of(r,HH[l])]=g[l,r]
    return out
print("inv1024 generic err", np.abs(inv1024_generic(Gs)-ref).max())
# ---------------- 2048 via even/odd 1024 transforms ----------------
u=rng.standard_normal(2048)
E,_=fwd1024(u[0::2].astype(complex)); O,_=fwd1024(u[1::2].astype(complex))
k=np.arange(1024)
U=E+np.exp(-2j*np.pi*k/2048)*O
ref2=np.fft.fft(u)
print("fwd2048 err", max(np.abs(U-ref2[:1024]).max(), abs((E[0]-O[0])-ref2[1024])))
G=np.zeros(2048,complex); G[:1025]=rng.standard_normal(1025)+1j*rng.standard_normal(1025)
want=np.real(np.fft.ifft(G)*2048)
got=np.zeros(2048)
for p in range(2):
    Gp=G[:1024]*np.exp(+2j*np.pi*k*p/2048)
    got[p::2]=inv1024_generic(Gp)+G[1024].real*(1 if p==0 else -1)
print("inv2048 err", np.abs(got-want).max())
