"""Generates the monomial tables of the 5-point solver (3 variables x,y,z, degree <= 3).
Printed once and pasted into oracle/fivepoint.py and pose_estimation_amd/csrc/fivepoint.hip."""
import itertools

def monos(deg):
    out = []
    for d in range(deg, -1, -1):          # graded: highest degree first
        for a in range(d, -1, -1):
            for b in range(d - a, -1, -1):
                out.append((a, b, d - a - b))
    return out

M1, M2, M3 = monos(1), monos(2), monos(3)
mul11 = [[M2.index(tuple(p + q for p, q in zip(a, b))) for b in M1] for a in M1]
mul21 = [[M3.index(tuple(p + q for p, q in zip(a, b))) for b in M1] for a in M2]
nister = [(3,0,0),(0,3,0),(2,1,0),(1,2,0),(2,0,1),(2,0,0),(0,2,1),(0,2,0),(1,1,1),(1,1,0),
          (1,0,2),(1,0,1),(1,0,0),(0,1,2),(0,1,1),(0,1,0),(0,0,3),(0,0,2),(0,0,1),(0,0,0)]
order = [M3.index(m) for m in nister]
print("M1 =", M1); print("M2 =", M2); print("M3 =", M3)
print("MUL11 =", mul11); print("MUL21 =", mul21); print("NISTER =", order)
