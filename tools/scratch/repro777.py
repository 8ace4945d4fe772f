import json, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import oracle_probaln
from oracle import orc
from secphase_amd import api
c = json.load(open(os.path.join(ROOT, "tools/scratch/fuzz777.json")))
ref = np.array([int(x) for x in c["ref"]], np.uint8); q = np.array([int(x) for x in c["qry"]], np.uint8)
ctx = api.Context(0)
st, qq, _ = ctx.probaln_batch([ref], [q], [c["set_q"]], [(c["d"], c["e"], c["bw"])])
_, est, eq = oracle_probaln(ref, q, c["set_q"], c["d"], c["e"], c["bw"])
bad = np.nonzero(np.asarray(st[0]) != np.asarray(est))[0]
print("rows with a different state:", len(bad), bad[:20])
print("q differs at:", np.nonzero(np.asarray(qq[0]) != np.asarray(eq))[0][:20])
sc, zM, zI = ctx.probaln_posteriors([ref], [q], [c["set_q"]], [(c["d"], c["e"], c["bw"])], which=0)
s, oM, oI = orc.probaln_posteriors(ref, q, c["set_q"], c["d"], c["e"], c["bw"])
L = len(q)
with np.errstate(all="ignore"):
    print("1/s equal:", np.array_equal(sc[1:L], 1.0 / s[1:L], equal_nan=True), "zM equal:", np.array_equal(zM, oM, equal_nan=True), "zI equal:", np.array_equal(zI, oI, equal_nan=True))
for i in bad[:6]:
    i = int(i)
    print("row", i, "device state", st[0][i], "oracle", est[i], "q", qq[0][i], eq[i])
    zr, orr = np.asarray(zM[i]), np.asarray(oM[i])
    k = np.nonzero(zr != orr)[0]
    print("   zM row differs at", k[:10], "device", zr[k[:4]], "oracle", orr[k[:4]], " max dev", np.nanmax(zr), np.nanargmax(zr), "max orc", np.nanmax(orr), np.nanargmax(orr))
    zr, orr = np.asarray(zI[i]), np.asarray(oI[i])
    print("   zI max dev", np.nanmax(zr), np.nanargmax(zr), "max orc", np.nanmax(orr), np.nanargmax(orr))
