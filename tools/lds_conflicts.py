"""Developer tool: LDS bank-conflict model of the A/B-fragment ds_read_b128 of the encoder / decoder kernels
(MI355X_MICROARCH.md, LDS section: 64 banks x 4 B, a ds_read_b128 is served in four groups of 16 lanes,
one LDS cycle per group when no two lanes of the group read different addresses in the same 16-byte slot
column).  Evaluates a pixel-swizzle s(xx, yy) over every (tile, tap, k-chunk) of a level's geometry."""
import itertools
import sys

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
          list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          [32 + x for x in list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))],
          [32 + x for x in list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]]


def cycles(addrs):
    """addrs[lane] = byte address of the lane's 16-byte read -> LDS cycles of the wave instruction."""
    tot = 0
    for g in GROUPS:
        slots = {}
        for l in g:
            slots.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
        tot += max(len(v) for v in slots.values())
    return tot


def enc_level(cin, W, Hp, Wp, rb, swz):
    """enc_mfma<cin, ...>: tile rows 2*rb+2, TC = W+2, pixel = cin*2 bytes, chunk' = (2kc+kh) ^ swz(xx,yy)."""
    TC, PS, KC, CPP = W + 2, cin * 2, cin // 16, cin // 8
    nwin = rb * Wp
    ntiles = (nwin + 7) // 8
    tot = ideal = 0
    for tile in range(ntiles):
        for ky, kx, kc in itertools.product(range(3), range(3), range(KC)):
            addrs = []
            for lane in range(64):
                m, kh = lane & 31, lane >> 5
                win = min(tile * 8 + (m >> 2), nwin - 1)
                wy, wx = divmod(win, Wp)
                yy, xx = 2 * wy + ((m >> 1) & 1) + ky, 2 * wx + (m & 1) + kx
                addrs.append((yy * TC + xx) * PS + (((kc * 2 + kh) ^ swz(xx, yy)) % CPP) * 16)
            tot += cycles(addrs)
            ideal += 4
    return tot, ideal


def dec_level(C, Hi, Wi, nu, swz):
    """dec_mfma: grid (nu rows) x (Wi+1), TC = Wi+2, pixel = C*2 bytes."""
    TC, PS, KC, CPP = Wi + 2, C * 2, C // 16, C // 8
    GW = Wi + 1
    npos = nu * GW
    tot = ideal = 0
    for tile in range((npos + 31) // 32):
        for a, b, kc in itertools.product(range(2), range(2), range(KC)):
            addrs = []
            for lane in range(64):
                q = min(tile * 32 + (lane & 31), npos - 1)
                ul, v = divmod(q, GW)
                yy, xx = ul + 1 - a, v + 1 - b
                addrs.append((yy * TC + xx) * PS + (((kc * 2 + (lane >> 5)) ^ swz(xx, yy)) % CPP) * 16)
            tot += cycles(addrs)
            ideal += 4
    return tot, ideal


if __name__ == "__main__":
    def cur_enc(cin):
        cpp = cin // 8
        xs = 16 // cpp
        return lambda xx, yy: ((xx // xs) % max(cpp // 2, 1)) | ((yy & 1) * (cpp // 2))
    def cur_dec(C):
        cpp = C // 8
        return lambda xx, yy: (xx // (16 // cpp)) % cpp
    print("current swizzles, 68x120 geometry (cycles / ideal):")
    for name, args in (("enc1", (16, 60, 17, 30, 3)), ("enc2", (32, 30, 8, 15, 4)), ("enc3", (64, 15, 4, 7, 4))):
        t, i = enc_level(*args, cur_enc(args[0]))
        print(f"  {name}: {t}/{i} = {t / i:.2f}x")
    for name, args in (("dec0", (128, 5, 8, 6)), ("dec1", (128, 9, 15, 5)), ("dec2", (64, 17, 30, 6)), ("dec3", (32, 34, 60, 35))):
        t, i = dec_level(*args, cur_dec(args[0]))
        print(f"  {name}: {t}/{i} = {t / i:.2f}x")


def search(kind, args, cpp):
    best = []
    fn = enc_level if kind == "enc" else dec_level
    for p, q in itertools.product(range(3), range(2)):
        for a, b, c in itertools.product(range(cpp), range(cpp), range(cpp)):
            swz = lambda xx, yy, p=p, q=q, a=a, b=b, c=c: ((xx >> p) * a + (yy >> q) * b + (yy & 1) * c) % cpp
            t, i = fn(*args, swz)
            best.append((t / i, (p, q, a, b, c)))
    best.sort()
    return best[:5]


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "search":
    for name, kind, args in (("enc2", "enc", (32, 30, 8, 15, 4)), ("enc3", "enc", (64, 15, 4, 7, 4)),
                             ("dec0", "dec", (128, 5, 8, 6)), ("dec2", "dec", (64, 17, 30, 6)), ("dec3", "dec", (32, 34, 60, 35)),
                             ("enc1", "enc", (16, 60, 17, 30, 3)), ("dec1", "dec", (128, 9, 15, 5))):
        print(name, search(kind, args, args[0] // 8))
