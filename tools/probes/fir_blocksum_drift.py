import numpy as np, sys
sys.path.insert(0, '.')
from fir_blocksum_proto import LEN, fc, LD
def run(x, b, T):
    N = LEN[b]; u_lo, u_hi = 728 - N // 2, 726 + N // 2; cL, cR = u_lo >> 5, u_hi >> 5
    w0 = 2 * LD(np.pi) * LD(fc(b)) / 48000; d = 2 * LD(np.pi) / N
    om = [w0, w0 + d, w0 - d]; g = [2 / LD(N), 1 / LD(N), 1 / LD(N)]; q = np.arange(32)
    rows3 = lambda c: [(g[i] * np.exp(-1j * (om[i] * (32 * c + q - 727)))).astype(np.complex128) for i in range(3)]
    def true_row(c, dt=np.complex128):
        m = 32 * c + q - 727
        win = np.where(np.abs(m) < N // 2, 4 / LD(N) * np.cos(LD(np.pi) * m / N) ** 2, 0)
        return (win * np.exp(-1j * (w0 * m))).astype(dt)
    X = x[:32 * (T + 50)].reshape(-1, 32)
    en, lv = rows3(cR - 1), rows3(cL)
    En = [X @ en[i] for i in range(3)]; Lv = [X @ lv[i] for i in range(3)]
    Re, Le = X @ true_row(cR), X @ true_row(cL)
    rot = [np.complex128(np.exp(1j * om[i] * 32)) for i in range(3)]
    J = cR - 1 - cL; V = [0j] * 3; out = np.zeros(T, np.complex128)
    for t in range(-J, T):
        for i in range(3):
            e = En[i][t + cR - 1] if t + cR - 1 >= 0 else 0
            l = Lv[i][t + cL] if t >= 0 else 0
            V[i] = rot[i] * V[i] + e - l
        if t >= 0: out[t] = V[0] + V[1] + V[2] + Re[t + cR] + Le[t + cL]
    # truth in long double for the last 200 outputs only
    m = np.arange(-(N // 2 - 1), N // 2)
    h = (4 / LD(N) * np.cos(LD(np.pi) * m / N) ** 2 * np.exp(-1j * (w0 * m))).astype(np.clongdouble)
    ts = list(range(0, 100)) + list(range(T - 100, T))
    ref = np.array([np.sum(h * x[32 * t + 727 + m].astype(LD)) for t in ts])
    sc = np.max(np.abs(ref))
    e = np.abs(out[ts] - ref) / sc
    return e[:100].max(), e[100:].max()
rng = np.random.default_rng(2)
T = 15000
x = rng.standard_normal(32 * (T + 60)) * 0.1
x2 = np.sin(2 * np.pi * fc(5) * np.arange(len(x)) / 48000)   # tone exactly at a band centre
for b in [0, 5, 15, 23]:
    print(b, "noise first/last 100:", run(x, b, T), " centre tone of band 5:", run(x2, b, T))
