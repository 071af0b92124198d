"""Resolution study, part 2: are the SkyTEM residuals a property of gatdaem1d's windows?
For every gate, fit TWO numbers shared by all soundings (a common shift of both window edges and a symmetric
widening, in samples of the digitising frequency) to the residual  reference / ours - 1  of 120 soundings (six earth
types x 20 wedge positions).  If the residual were model error, two numbers per gate could not explain 120 soundings;
they do (rms 1-4 permil -> 0.02-0.3 permil on the gates with signal), so the reference's values carry a per-gate,
sub-sample (|shift| <= 0.16 sample) window placement of their own.  Run: python scripts/tdem_study/window_jitter.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from evaluate import *
np.set_printoptions(linewidth=250, precision=3, suppress=True)
SK=(-13.0,0.0,2.0)
class Sys2(System):
    def series(self,sig,thk):
        r=np.hypot(self.dx,self.dy); a=self.a; H=2*self.alt+self.dz; l0,l1=base_abscissae(); lam,w=l0/r,W0_J0_120
        src=lam*j1(lam*a)/(2*np.pi*a); k=np.exp(-lam*H)*src*w/r
        Hn=np.array([np.sum(to.rte(lam,2*np.pi*f,sig,thk)*k) for f in self.fn])
        return Hn
for name,cols in [("SkytemHM.stm",slice(15,41)),("SkytemLM.stm",slice(41,60))]:
    S=Sys2(name,SK,30.0,start=1)
    stm=S.stm; N=S.N; fs=S.fs; f0=S.f0
    # rebuild time series operator
    wt, wc = stm["wave"][:, 0], stm["wave"][:, 1]; t = wt[0] + np.arange(N) / fs
    c = np.interp(t[: N // 2], wt, wc); cur = np.concatenate([c, -c]); I=np.fft.rfft(cur); fk=np.arange(N//2+1)*f0
    fac=np.full(fk.size,MU0,dtype=complex)*(-2j*np.pi*fk)
    for fc,n_ in zip(stm["CutOffFrequency"].split(), stm["Order"].split()): fac*=(1/(1+1j*fk/float(fc)))**int(float(n_))
    fac[0]=0; x=np.log10(S.fn); lf=np.log10(np.clip(fk[1:],S.fn[0],S.fn[-1]))
    data=[]
    for model in sorted(WEDGE_CONDUCTIVITY):
        sk=np.loadtxt(os.path.join(GOLDEN,f"skytem_{model}_clean.csv"),delimiter=",",skiprows=1)
        for i in range(0,79,4):
            Hn=S.series(WEDGE_CONDUCTIVITY[model],[ZW[i],ZD[i]-ZW[i]])
            Hk=np.zeros(fk.size,complex); Hk[1:]=CubicSpline(x,Hn.real,bc_type='natural')(lf)+1j*CubicSpline(x,Hn.imag,bc_type='natural')(lf)
            rr=np.fft.irfft(I*fac*Hk,N); ref=sk[i,cols]; st=[]
            for a_,b_ in stm['windows']:
                q=np.linspace(a_,b_,2049); f=np.interp(q,t,rr); m=np.trapezoid(f,q)/(b_-a_)
                st.append((m,(m-f[0])/(b_-a_),(f[-1]-m)/(b_-a_)))
            data.append((ref,np.array(st)))
    print(name,'dt=%.1f ns'%(1e9/fs))
    for k in range(len(stm['windows'])):
        ref=np.array([d[0][k] for d in data]); m=np.array([d[1][k,0] for d in data]); da=np.array([d[1][k,1] for d in data]); db=np.array([d[1][k,2] for d in data])
        amp=np.array([abs(d[0][k])/np.abs(d[0]).max() for d in data]); ok=amp>3e-3
        if ok.sum()<10: continue
        y=(ref/m-1)[ok]; Xs=((da+db)/m)[ok]; Xw=((db-da)/m)[ok]     # shift (both edges +), widen (a-, b+)
        X=np.c_[Xs,Xw]; sol=np.linalg.lstsq(X,y,rcond=None)[0]; r2=y-X@sol
        cov=np.linalg.inv(X.T@X)*np.var(r2); se=np.sqrt(np.diag(cov))
        a_,b_=stm['windows'][k]
        print('%2d n=%3d rms %.2f->%.2f‰  shift %+7.3f±%.3f samp  widen(each side) %+7.3f±%.3f samp | a*fs frac %.3f b*fs frac %.3f width %.2f'%(k+1,ok.sum(),1e3*np.sqrt(np.mean(y**2)),1e3*np.sqrt(np.mean(r2**2)),sol[0]*fs,se[0]*fs,sol[1]*fs,se[1]*fs,((a_-t[0])*fs)%1,((b_-t[0])*fs)%1,(b_-a_)*fs))
