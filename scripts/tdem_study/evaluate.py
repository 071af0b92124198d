"""Resolution study: which discretisation of the GA-AEM pipeline reproduces the reference's TDEM known answers
(tests/golden/skytem_*_clean.csv, tempest_*_clean.csv)?  numpy only; run: python scripts/tdem_study/evaluate.py"""
import json, os, sys
import numpy as np
from scipy.interpolate import CubicSpline
from scipy.special import j1
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import tdem_oracle as to
from oracle.gs_filters import W0_J0_120, W1_J1_140, base_abscissae
from conftest import GOLDEN, WEDGE_CONDUCTIVITY
MU0 = 4e-7 * np.pi
ZW, ZD = np.linspace(50.0, 1.0, 79), np.linspace(75.0, 500.0, 79)


class System:
    def __init__(self, name, offset, alt, fpd=None, start=0, tol=0.0, dense=False):
        stm = self.stm = to.parse_stm(os.path.join(GOLDEN, name))
        self.f0, self.fs = float(stm["BaseFrequency"]), float(stm["WaveformDigitisingFrequency"])
        self.N = N = int(round(self.fs / self.f0))
        self.dx, self.dy, self.dz = offset; self.alt = alt
        self.a = float(stm.get("ModellingLoopRadius", 0.0))
        fpd = float(stm.get("FrequenciesPerDecade", 5)) if fpd is None else fpd
        if dense:
            self.fn = 10 ** np.arange(np.log10(self.f0) - 0.5, np.log10(self.fs / 2) + 0.2, 1 / 12.0)
        else:
            n = int(np.ceil((np.log10(self.fs / 2) - np.log10(self.f0)) * fpd)) + 1
            self.fn = self.f0 * 10 ** ((np.arange(n + start) - start) / fpd)
        wt, wc = stm["wave"][:, 0], stm["wave"][:, 1]
        t = wt[0] + np.arange(N) / self.fs
        if abs((wt[-1] - wt[0]) - 0.5 / self.f0) <= 2.0 / self.fs:
            c = np.interp(t[: N // 2], wt, wc); cur = np.concatenate([c, -c])
        else:
            cur = np.interp(t, wt, wc)
        I = np.fft.rfft(cur); fk = np.arange(N // 2 + 1) * self.f0
        moment = float(stm.get("NumberOfTurns", 1)) * float(stm.get("PeakCurrent", 1)) * float(stm.get("LoopArea", 1))
        fac = np.full(fk.size, MU0 * moment, dtype=complex)
        if stm.get("OutputType", "dB/dt").lower().startswith("db"):
            fac *= -2j * np.pi * fk
        if "CutOffFrequency" in stm:
            for fc, n_ in zip(stm["CutOffFrequency"].split(), stm["Order"].split()):
                fac *= (1.0 / (1.0 + 1j * fk / float(fc))) ** int(float(n_))
        fac[0] = 0
        nw = len(stm["windows"]); A = np.zeros((nw, N))
        area = stm.get("WindowWeightingScheme", "Boxcar").lower().startswith("area")
        for w, (a_, b_) in enumerate(stm["windows"]):
            if area:
                q = np.linspace(a_, b_, 1025); wq = np.full(q.size, (b_ - a_) / (q.size - 1)); wq[0] *= .5; wq[-1] *= .5
                pos = (q - t[0]) * self.fs; i0 = np.floor(pos).astype(int); fr = pos - i0
                np.add.at(A[w], i0, wq * (1 - fr) / (b_ - a_)); np.add.at(A[w], i0 + 1, wq * fr / (b_ - a_))
            else:
                m = (t >= a_ - tol) & (t <= b_ + tol); A[w, m] = 1.0 / m.sum()
        x = np.log10(self.fn); S = np.zeros((fk.size, x.size))
        S[1:] = CubicSpline(x, np.eye(x.size), bc_type="natural")(np.log10(np.clip(fk[1:], self.fn[0], self.fn[-1])))
        G = (I * fac)[:, None] * S
        self.Wre = (A @ np.fft.irfft(G, N, axis=0)); self.Wim = (A @ np.fft.irfft(1j * G, N, axis=0))
        self.comps = [c for c in "XZ" if float(stm.get(c + "OutputScaling", 0.0)) != 0.0]
        self.scale = {c: float(stm.get(c + "OutputScaling", 0.0)) for c in "XZ"}

    def forward(self, sig, thk):
        r = np.hypot(self.dx, self.dy); a = self.a; H = 2 * self.alt + self.dz
        l0, l1 = base_abscissae(); out = []
        for comp in self.comps:
            lam, w = (l0 / r, W0_J0_120) if comp == "Z" else (l1 / r, W1_J1_140)
            src = lam * j1(lam * a) / (2 * np.pi * a) if a > 0 else lam ** 2 / (4 * np.pi)
            k = np.exp(-lam * H) * src * w / r * (1.0 if comp == "Z" else -self.dx / r)
            Hn = np.array([np.sum(to.rte(lam, 2 * np.pi * f, sig, thk) * k) for f in self.fn]) * self.scale[comp]
            out.append(self.Wre @ Hn.real + self.Wim @ Hn.imag)
        return np.concatenate(out)


def stats(systems, family, cols, step=1):
    """relative error by amplitude class over all six earth types x rows"""
    rel, amp, absn = [], [], []
    for model in sorted(WEDGE_CONDUCTIVITY):
        ref_all = np.loadtxt(os.path.join(GOLDEN, f"{family}_{model}_clean.csv"), delimiter=",", skiprows=1)
        for i in range(0, 79, step):
            sig, thk = WEDGE_CONDUCTIVITY[model], [ZW[i], ZD[i] - ZW[i]]
            for S, (c0, c1, nper) in zip(systems, cols):
                v = S.forward(sig, thk); ref = ref_all[i, c0:c1]
                for j in range(0, ref.size, nper):          # per component block
                    rr, vv = ref[j:j + nper], v[j:j + nper]; pk = np.abs(rr).max()
                    rel.append(vv / rr - 1); amp.append(np.abs(rr) / pk); absn.append((vv - rr) / pk)
    rel, amp, absn = map(np.concatenate, (rel, amp, absn))
    out = {}
    for lo, hi in [(1e-1, 2), (1e-2, 1e-1), (1e-3, 1e-2), (1e-4, 1e-3), (0, 1e-4)]:
        m = (amp >= lo) & (amp < hi)
        if m.any():
            out[f"amp>={lo:g}"] = dict(n=int(m.sum()), max_rel=float(np.abs(rel[m]).max()), median_rel=float(np.median(np.abs(rel[m]))),
                                       max_abs_over_peak=float(np.abs(absn[m]).max()))
    out["all"] = dict(max_abs_over_peak=float(np.abs(absn).max()))
    return out


if __name__ == "__main__":
    step = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    SK, TP = (-13.0, 0.0, 2.0), (-107.0, 0.0, -45.0)
    results = {}
    for tag, kw in [("dense nodes (12/decade from 0.3 f0), strict boxcar", dict(dense=True)),
                    ("stm FrequenciesPerDecade from f0, strict boxcar", dict()),
                    ("stm FrequenciesPerDecade from f0, boxcar +-1e-7 s", dict(tol=1e-7)),
                    ("stm fpd, one node below f0, boxcar +-1e-7 s", dict(tol=1e-7, start=1)),
                    ("stm fpd, two nodes below f0, boxcar +-1e-7 s", dict(tol=1e-7, start=2)),
                    ("stm fpd, five nodes below f0, boxcar +-1e-7 s", dict(tol=1e-7, start=5))]:
        hm, lm = System("SkytemHM.stm", SK, 30.0, **kw), System("SkytemLM.stm", SK, 30.0, **kw)
        te = System("tempest.stm", TP, 120.0, **kw)
        results[tag] = dict(skytem=stats([hm, lm], "skytem", [(15, 41, 26), (41, 60, 19)], step),
                            tempest=stats([te], "tempest", [(17, 47, 15)], step))
        print(tag); print(json.dumps(results[tag], indent=1))
    json.dump(results, open(os.path.join(HERE, "results.json"), "w"), indent=1)
