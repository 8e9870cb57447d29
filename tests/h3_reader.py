"""Reader for HMMER's pressed `.h3f` / `.h3p` files (test infrastructure).

Layout restated from SURVEY.md section 8(f) (upstream impl_sse/io.c p7_oprofile_Write); used to pin the
profile conversion bit-for-bit against the fixtures pressed by real HMMER 3.3.1
(reference src/pyhmmer/tests/data/README.md:16-25).
"""
import struct
import numpy as np


def _Q(M, w):
    return max(2, (M - 1) // w + 1)


class Reader:
    def __init__(self, data):
        self.d, self.p = data, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.d, self.p)
        self.p += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def arr(self, dtype, n):
        a = np.frombuffer(self.d, dtype=dtype, count=n, offset=self.p).copy()
        self.p += a.nbytes
        return a

    def string(self):
        n = self.take("i")
        if n <= 0:
            return None
        s = self.d[self.p:self.p + n].decode()
        self.p += n + 1
        return s

    def eof(self):
        return self.p >= len(self.d)


def read_h3f(path):
    r = Reader(open(path, "rb").read())
    out = []
    while not r.eof():
        magic = r.take("I")
        assert magic == 0xb3e6e6f3, hex(magic)
        M, abc = r.take("ii")
        Kp = 29 if abc == 3 else 18
        n = r.take("i")
        name = r.d[r.p:r.p + n].decode(); r.p += n + 1
        max_length = r.take("i")
        tbm, tec, tjb = r.take("BBB")
        scale_b = r.take("f")
        base, bias = r.take("BB")
        Q = _Q(M, 16)
        sbv = r.arr(np.int8, Kp * (Q + 17) * 16).reshape(Kp, (Q + 17) * 16)
        rbv = r.arr(np.uint8, Kp * Q * 16).reshape(Kp, Q * 16)
        evparam = r.arr(np.float32, 6)
        offs = r.arr(np.int64, 3)
        compo = r.arr(np.float32, 20)
        assert r.take("I") == 0xb3e6e6f3
        out.append(dict(M=M, abc=abc, name=name, max_length=max_length, tbm=tbm, tec=tec, tjb=tjb,
                        scale_b=scale_b, base=base, bias=bias, sbv=sbv, rbv=rbv, evparam=evparam,
                        offs=offs, compo=compo))
    return out


def read_h3p(path):
    r = Reader(open(path, "rb").read())
    out = []
    while not r.eof():
        magic = r.take("I")
        assert magic == 0xb3e6f0f3, hex(magic)
        M, abc = r.take("ii")
        Kp = 29 if abc == 3 else 18
        name, acc, desc = r.string(), r.string(), r.string()
        rf = r.d[r.p:r.p + M + 2]; r.p += M + 2
        mm = r.d[r.p:r.p + M + 2]; r.p += M + 2
        cs = r.d[r.p:r.p + M + 2]; r.p += M + 2
        cons = r.d[r.p:r.p + M + 2]; r.p += M + 2
        Q8, Q4 = _Q(M, 8), _Q(M, 4)
        twv = r.arr(np.int16, 8 * Q8 * 8).reshape(8 * Q8, 8)
        rwv = r.arr(np.int16, Kp * Q8 * 8).reshape(Kp, Q8 * 8)
        xw = r.arr(np.int16, 8).reshape(4, 2)
        scale_w = r.take("f")
        base_w, ddbound_w = r.take("hh")
        ncj_roundoff = r.take("f")
        tfv = r.arr(np.float32, 8 * Q4 * 4).reshape(8 * Q4, 4)
        rfv = r.arr(np.float32, Kp * Q4 * 4).reshape(Kp, Q4 * 4)
        xf = r.arr(np.float32, 8).reshape(4, 2)
        cutoff = r.arr(np.float32, 6)
        nj = r.take("f")
        mode, L = r.take("ii")
        assert r.take("I") == 0xb3e6f0f3
        out.append(dict(M=M, abc=abc, name=name, acc=acc, desc=desc, consensus=cons, twv=twv, rwv=rwv, xw=xw,
                        scale_w=scale_w, base_w=base_w, ddbound_w=ddbound_w, ncj_roundoff=ncj_roundoff,
                        tfv=tfv, rfv=rfv, xf=xf, cutoff=cutoff, nj=nj, mode=mode, L=L))
    return out
