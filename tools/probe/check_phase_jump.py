"""CPU check of the phase fast-forward of csrc/phaser.hip:ps_phase_after (numpy float32 = the fp32 arithmetic of the
kernel): g sequential  p <- fl(p + inc); wrap at 2 pi  steps against the O(binades) jump algorithm, random LFO rates and
constructed round-half-even ties.      python tools/probe/check_phase_jump.py"""
import numpy as np, math, random
f32=np.float32
TWO_PI=f32(6.283185307179586476925286766559)
def seq(inc, g):
    p=f32(0)
    for _ in range(g):
        p=f32(p+inc)
        while p>=TWO_PI: p=f32(p-TWO_PI)
    return p
def step(p,inc):
    p=f32(p+inc)
    if p>=TWO_PI: p=f32(p-TWO_PI)
    return p
def expo(x):
    return math.frexp(float(x))[1]
def jump(inc, g):
    p=f32(0); rem=g; iters=0
    while rem>0:
        iters+=1
        p0=p
        p1=step(p0,inc); rem-=1
        if rem==0: p=p1; break
        p2=step(p1,inc); rem-=1
        if rem==0: p=p2; break
        p3=step(p2,inc); rem-=1
        p=p3
        if rem==0: break
        d1=f32(p2-p1); d2=f32(p3-p2)
        if d1==d2 and d2>0 and expo(p1)==expo(p2)==expo(p3):
            lim=min(float(np.ldexp(1.0,expo(p3))), float(TWO_PI))
            room=lim-float(p3)
            j=int(math.floor(room/float(d2)))-1
            if j>rem: j=rem
            if j>0:
                p=f32(float(p3)+j*float(d2)); rem-=j
    return p, iters
random.seed(1)
bad=0; maxit=0
for t in range(400):
    rate=f32(math.exp(random.uniform(math.log(0.5),math.log(3.0))))
    if t%7==0: rate=f32(random.choice([0.5,1.0,2.0,3.0,0.75,1.5]))
    inc=f32(f32(TWO_PI/f32(44100.0/4.0))*rate)
    g=random.randint(0,44100)
    a=seq(inc,g); b,it=jump(inc,g); maxit=max(maxit,it)
    if a!=b: bad+=1; print("MISMATCH",rate,g,a,b)
print("bad",bad,"maxit",maxit)
# adversarial: inc values with tie fractions in some binade
for t in range(200):
    e=random.randint(-3,2)  # binade exponent of p
    ulp=2.0**(e-23)
    I=random.randint(100,4000)
    inc=f32((I+0.5)*ulp)
    g=random.randint(1000,30000)
    a=seq(inc,g); b,it=jump(inc,g)
    if a!=b: bad+=1; print("TIE MISMATCH",inc,g,a,b)
print("bad after ties",bad)
