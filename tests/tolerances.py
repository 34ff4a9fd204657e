"""Per-stage parity bounds shared by the GPU graph / golden tests (bf16 activations between kernels, fp32 accumulation)."""
# relative L2 error of each stage output on the oracle's inputs: measured (20x36 / 400x600) -> bound (<= 2x)
TOL = {"cond_feat": 4.5e-3,    # 2.1e-3 / 2.1e-3
       "color_map": 2.0e-2,    # 9.7e-3 / 9.1e-3
       "mid_feat0": 6.5e-3,    # 3.1e-3 / 3.1e-3
       "mid_feat1": 1.25e-2,   # 6.1e-3 / 6.2e-3
       "latent": 5.5e-3,       # 2.8e-3 / 2.4e-3   (flow reverse on the oracle's color_map / cond_feat)
       "code_feat0": 2.1e-2,   # 1.03e-2 / 1.04e-2
       "code_feat1": 3.4e-2,   # 1.60e-2 / 1.67e-2
       "vq_rec": 3.6e-2,       # 1.61e-2 / 1.76e-2
       "aft_out": 2.4e-2}      # 1.21e-2 / 0.82e-2
