import numpy as np
rng = np.random.default_rng(0)

def bf16_rne(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
    return r.view(np.float32)

def mm32(a, b):
    return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float32)

def mm_bf3(A, B):
    def split(x):
        h = bf16_rne(x); r = x - h; m = bf16_rne(r); r = r - m; l = bf16_rne(r); return h, m, l
    Ah, Am, Al = split(A); Bh, Bm, Bl = split(B)
    acc = mm32(Ah, Bl) + mm32(Al, Bh) + mm32(Am, Bm)
    acc = acc + mm32(Am, Bh) + mm32(Ah, Bm)
    return (acc + mm32(Ah, Bh)).astype(np.float32)

def f16split(x, e):
    # x * 2^e into two fp16 pieces
    xs = np.ldexp(x.astype(np.float32), e)
    h = xs.astype(np.float16).astype(np.float32)
    m = (xs - h).astype(np.float16).astype(np.float32)
    return h, m

def scale_exp(x, target=14):
    mx = float(np.abs(x).max())
    if mx == 0: return 0
    return target - int(np.floor(np.log2(mx))) - 1   # max*2^e in [2^(target-1), 2^target)

def mm_f16(A, B, nterms=3):
    ea, eb = scale_exp(A), scale_exp(B)
    Ah, Am = f16split(A, ea); Bh, Bm = f16split(B, eb)
    acc = mm32(Am, Bh) + mm32(Ah, Bm)
    if nterms == 4: acc = acc + mm32(Am, Bm)
    acc = acc + mm32(Ah, Bh)
    return np.ldexp(acc.astype(np.float32), -(ea + eb)).astype(np.float32)

def power(M, ex, mm):
    P = M.copy(); hb = ex.bit_length() - 1
    for b in range(hb - 1, -1, -1):
        P = mm(P, P)
        if (ex >> b) & 1: P = mm(M, P)
    return P

def trial(d, alpha, sscale, seed):
    r = np.random.default_rng(seed)
    s = sscale * r.standard_normal((d, d))
    u = r.uniform(1e-7, 1, (d, d))
    eps = np.log(u / (1 - u))
    g = 1 / (1 + np.exp(-(eps + alpha * s))); np.fill_diagonal(g, 0)
    M64 = np.eye(d) + g / d
    ref = np.linalg.matrix_power(M64, d - 1)
    M32 = M64.astype(np.float32)
    fac = alpha * g * (1 - g)
    out = {}
    for name, mm in (("f32", mm32), ("bf16x3", mm_bf3), ("f16x2", lambda a, b: mm_f16(a, b, 3)), ("f16x2+mm", lambda a, b: mm_f16(a, b, 4))):
        P = power(M32, d - 1, mm).astype(np.float64)
        W, Wr = P.T * fac, ref.T * fac
        emax = np.abs(P - ref).max() / np.abs(ref).max()
        nz = ref > 0
        eel = (np.abs(P - ref)[nz] / ref[nz]).max()
        ew = np.abs(W - Wr).max() / max(np.abs(Wr).max(), 1e-300)
        out[name] = (emax, eel, ew)
    return out

for (d, alpha, ss) in ((50, 0.0, 1), (50, 0.5, 1), (50, 5, 1), (50, 20, 1), (50, 300, 1), (50, 1000, 1), (64, 2, 1), (40, 10, 1), (50, 50, 0.2)):
    acc = {}
    for seed in range(6):
        o = trial(d, alpha, ss, seed)
        for k, v in o.items():
            acc.setdefault(k, []).append(v)
    print(f"d={d} alpha={alpha} sscale={ss}")
    for k, v in acc.items():
        v = np.array(v)
        print(f"   {k:9s} max-rel P {v[:,0].max():.2e}  elementwise P {v[:,1].max():.2e}  max-rel W {v[:,2].max():.2e}")
