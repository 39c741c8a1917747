import mpmath as mp, numpy as np
mp.mp.dps=60
def cheb_fit(f, lo, hi, deg):
    n=deg+1
    nodes=[mp.cos(mp.pi*(k+mp.mpf(1)/2)/n) for k in range(n)]
    xs=[(hi-lo)/2*t+(hi+lo)/2 for t in nodes]
    ys=[f(x) for x in xs]
    # chebyshev coefficients
    c=[]
    for j in range(n):
        s=mp.fsum(ys[k]*mp.cos(mp.pi*j*(k+mp.mpf(1)/2)/n) for k in range(n))
        c.append(2*s/n)
    c[0]/=2
    # convert to monomial in x: T_j(t), t=(2x-(hi+lo))/(hi-lo)
    # build polynomial via recurrence with mp coefficients
    def padd(a,b):
        m=max(len(a),len(b)); return [(a[i] if i<len(a) else 0)+(b[i] if i<len(b) else 0) for i in range(m)]
    def pmul(a,b):
        r=[mp.mpf(0)]*(len(a)+len(b)-1)
        for i,x in enumerate(a):
            for j,y in enumerate(b): r[i+j]+=x*y
        return r
    tpoly=[-(hi+lo)/(hi-lo), 2/(hi-lo)]
    T=[[mp.mpf(1)], tpoly]
    for j in range(2,n):
        T.append(padd(pmul([mp.mpf(0)] ,[0]), padd([2*x for x in pmul(tpoly,T[j-1])], [-x for x in T[j-2]])))
    res=[mp.mpf(0)]
    for j in range(n): res=padd(res,[c[j]*x for x in T[j]])
    return res
def P(z):
    if z==0: return mp.mpf(1)/6
    s=mp.sqrt(z); return (mp.asin(s)/s-1)/z
for deg in (11,12,13,14):
    co=cheb_fit(P, mp.mpf(0), mp.mpf(1)/4, deg)
    cod=[float(x) for x in co]
    # max error of polynomial (double coefficients, exact evaluation) on grid
    err=0
    for i in range(0,2001):
        z=mp.mpf(i)/8000
        v=mp.fsum(mp.mpf(cod[k])*z**k for k in range(len(cod)))
        err=max(err, abs(v-P(z)))
    print(deg, mp.nstr(err,5))
    if deg==11: best=cod
print("asinP =", [repr(x) for x in best])
# cos taylor-like minimax on y^2 in [0, 1.2]: C(w) = (cos(sqrt w)-1+w/2)/w^2
def C(w):
    if w==0: return mp.mpf(1)/24
    y=mp.sqrt(w); return (mp.cos(y)-1+w/2)/(w*w)
for deg in (6,7,8):
    co=cheb_fit(C, mp.mpf(0), mp.mpf('1.21'), deg)
    cod=[float(x) for x in co]
    err=0
    for i in range(0,1211):
        w=mp.mpf(i)/1000
        v=mp.fsum(mp.mpf(cod[k])*w**k for k in range(len(cod)))
        err=max(err, abs(v-C(w))*w*w)
    print('cos',deg, mp.nstr(err,5))
    if deg==6: bestc=cod
print("cosC =", [repr(x) for x in bestc])
