import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2]); w = int(sys.argv[3])
a = a.reshape(-1, w); b = b.reshape(-1, w)
bad = np.nonzero((a != b).any(axis=1))[0]
print("differing units:", len(bad), bad[:20].tolist())
for i in bad[:6]:
    print(i, a[i].view(np.uint32)[:10].tolist(), "|", b[i].view(np.uint32)[:10].tolist())
