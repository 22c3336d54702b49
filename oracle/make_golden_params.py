"""Scene parameters shared by oracle/make_golden.py (which needs /root/reference) and the tests that recompute the oracle's
side of a fixture (which do not)."""
# tests/scenes.py::ba_pin_scene arguments of tests/golden/ba_f64_pin.npz: windows of 10 and 30 free poses
BA_PIN = {"w10": dict(seed=33, n_frames=11, M=8, lifetime=4), "w30": dict(seed=32, n_frames=31, M=8, lifetime=4)}
