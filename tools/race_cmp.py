import sys, torch
la, lb = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for t in range(la.shape[0]):
    if not (torch.equal(la[t, 2], lb[t, 2]) and torch.equal(la[t, 8], lb[t, 8])) and not (torch.isnan(la[t,2]) and torch.isnan(lb[t,2])):
        print("first difference at frame", t)
        for u in range(max(0, t - 2), min(la.shape[0], t + 3)):
            print("  t=%d A net=%.12g w=%.12g E=%d | B net=%.12g w=%.12g E=%d" % (u, la[u][2], la[u][8], la[u][9], lb[u][2], lb[u][8], lb[u][9]))
        break
else:
    print("identical over", la.shape[0], "frames")
