import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import mrphy_amd, bloch_oracle as O
from mrphy_amd import beffective, sims, fused
dev = torch.device('cuda:0')
def run(N, nM, nT, dt_, relax=True, df=True):
    gen = torch.Generator().manual_seed(3)
    rnd = lambda *s: torch.rand(s, generator=gen, dtype=torch.float64)
    M0 = rnd(N, nM, 3).to(dt_); rf = (rnd(N,2,nT)*2-1).to(dt_); gr = (rnd(N,3,nT)*2-1).to(dt_)
    loc = ((rnd(N,nM,3)*2-1)*6).to(dt_); Df = ((rnd(N,nM)*2-1)*200).to(dt_) if df else None
    T1 = (0.5+rnd(N,nM)).to(dt_) if relax else None; T2 = (0.02+0.1*rnd(N,nM)).to(dt_) if relax else None
    γ = torch.tensor(4257.6, dtype=dt_); dt = torch.tensor([4e-6], dtype=dt_)
    bo = O.rfgr2beff(rf, gr, loc, Δf=Df, γ=γ)
    Mo = O.blochsim(M0, bo, T1=T1, T2=T2, γ=γ, dt=dt)
    d = lambda x: None if x is None else x.to(dev)
    with mrphy_amd.constants_on('cpu'):
        bh = beffective.rfgr2beff(d(rf), d(gr), d(loc), Δf=d(Df), γ=d(γ))
        M1 = sims.blochsim(d(M0), bh, T1=d(T1), T2=d(T2), γ=d(γ), dt=d(dt))
        M1o = sims.blochsim(d(M0), d(bo), T1=d(T1), T2=d(T2), γ=d(γ), dt=d(dt))
        M2 = fused.blochsim_rfgr(d(M0), d(rf), d(gr), d(loc), Δf=d(Df), γ_beff=d(γ), T1=d(T1), T2=d(T2), γ=d(γ), dt=d(dt))
    mx = lambda a, b: float((a.double().cpu()-b.double().cpu()).abs().max())
    print(f'N{N} nM{nM} nT{nT} {str(dt_)[-3:]} relax={relax} df={df}: beff {mx(bh,bo):.1e}  K1-ora {mx(M1,Mo):.1e}  K1(ora beff)-ora {mx(M1o,Mo):.1e}  K2-ora {mx(M2,Mo):.1e}  K2-K1 {mx(M2,M1):.1e}', flush=True)
for dt_ in (torch.float64, torch.float32):
    for (N,nM,nT) in ((1,63,17),(1,63,16),(1,63,1),(1,64,32),(1,1,4),(2,64,13)):
        run(N,nM,nT,dt_)
    run(1,63,17,dt_,relax=False)
    run(1,63,17,dt_,df=False)
    run(1,63,17,dt_,relax=False,df=False)
