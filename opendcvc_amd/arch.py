"""Parameter inventory of the two DCVC-RT codecs (names + shapes), torch-free.

The names are the reference checkpoints' state_dict keys, so a reference ``*.pth.tar`` loads
unchanged (drop-in seam 1).  Architecture facts restated from:
  DMC   /root/reference/src/models/video_model.py:15-22,25-247
  DMCI  /root/reference/src/models/image_model.py:13-141
  DepthConvBlock / SubpelConv2x / ResidualBlock*  /root/reference/src/layers/layers.py:29-156
  BitEstimator / Bitparm  /root/reference/src/models/entropy_models.py:84-137
"""

QP_NUM = 64
DMC_QP_SHIFT = (0, 8, 4)
DMC_EXTRA_QP = max(DMC_QP_SHIFT)

CH_SRC = 3 * 8 * 8      # pixel-unshuffled YCbCr444 frame
DMC_CH_D = 256          # feature width
DMC_CH_Y = 128
DMC_CH_Z = 128
DMC_CH_RECON = 320
DMCI_CH = 368           # intra trunk width
DMCI_N = 256            # intra latent channels
DMCI_CH_Z = 128


class Spec:
    """Ordered list of (name, shape, kind).  kind in {'w','b','dw','q','bitparm'} selects the
    initialiser of the synthetic generator (weights.py)."""

    def __init__(self):
        self.items = []

    def add(self, name, shape, kind):
        self.items.append((name, tuple(int(s) for s in shape), kind))

    # -- layer helpers -------------------------------------------------------
    def conv(self, prefix, cin, cout, k=1, last_in_branch=False):
        self.add(prefix + ".weight", (cout, cin, k, k), "w_res" if last_in_branch else "w")
        self.add(prefix + ".bias", (cout,), "b")

    def dcb(self, prefix, cin, c, force_adaptor=False):
        if cin != c or force_adaptor:
            self.conv(prefix + ".adaptor", cin, c)
        self.conv(prefix + ".dc.0", c, c)
        self.add(prefix + ".dc.2.weight", (c, 1, 3, 3), "dw")
        self.add(prefix + ".dc.2.bias", (c,), "b")
        self.conv(prefix + ".dc.3", c, c, last_in_branch=True)
        self.conv(prefix + ".ffn.0", c, 4 * c)
        self.conv(prefix + ".ffn.2", 2 * c, c, last_in_branch=True)

    def res_down(self, prefix, cin, c):      # ResidualBlockWithStride2
        self.conv(prefix + ".down", cin, c, k=2)
        self.dcb(prefix + ".conv", c, c)

    def res_up(self, prefix, cin, c):        # ResidualBlockUpsample
        self.conv(prefix + ".up.conv.0", cin, 4 * c)
        self.dcb(prefix + ".conv", c, c)

    def bit_estimator(self, prefix, qp_num, ch):
        for f in ("f1", "f2", "f3"):
            for p in ("h", "b", "a"):
                self.add(f"{prefix}.{f}.{p}", (qp_num, ch, 1, 1), "bitparm")
        for p in ("h", "b"):
            self.add(f"{prefix}.f4.{p}", (qp_num, ch, 1, 1), "bitparm")


def dmc_spec():
    s = Spec()
    d, y, z, r = DMC_CH_D, DMC_CH_Y, DMC_CH_Z, DMC_CH_RECON
    nq = QP_NUM + DMC_EXTRA_QP
    for n in ("q_encoder", "q_decoder", "q_feature"):
        s.add(n, (nq, d, 1, 1), "q")
    s.add("q_recon", (nq, r, 1, 1), "q")
    s.bit_estimator("bit_estimator_z", nq, z)
    s.dcb("feature_adaptor_i", CH_SRC, d)
    s.conv("feature_adaptor_p", d, d)
    for i in range(2):
        s.dcb(f"feature_extractor.conv1.{i}", d, d)
    for i in range(4):
        s.dcb(f"feature_extractor.conv2.{i}", d, d)
    s.conv("encoder.conv1", CH_SRC, d)
    s.dcb("encoder.conv2.0", 2 * d, d)
    s.dcb("encoder.conv2.1", d, d)
    s.dcb("encoder.conv3", d, d)
    s.conv("encoder.down", d, y, k=3)
    s.dcb("hyper_encoder.conv.0", y, z)
    s.res_down("hyper_encoder.conv.1", z, z)
    s.res_down("hyper_encoder.conv.2", z, z)
    s.res_up("hyper_decoder.conv.0", z, z)
    s.res_up("hyper_decoder.conv.1", z, z)
    s.dcb("hyper_decoder.conv.2", z, y)
    s.res_down("temporal_prior_encoder", d, 2 * y)
    for i in range(3):
        s.dcb(f"y_prior_fusion.conv.{i}", 3 * y, 3 * y)
    s.conv("y_prior_fusion.conv.3", 3 * y, 3 * y)
    s.dcb("y_spatial_prior.conv.0", 4 * y, 3 * y)
    s.dcb("y_spatial_prior.conv.1", 3 * y, 3 * y)
    s.conv("y_spatial_prior.conv.2", 3 * y, 2 * y)
    s.conv("decoder.up.conv.0", y, 4 * d, k=3)
    s.dcb("decoder.conv1.0", 2 * d, d)
    s.dcb("decoder.conv1.1", d, d)
    s.dcb("decoder.conv1.2", d, d)
    s.conv("decoder.conv2", d, d)
    s.dcb("recon_generation_net.conv.0", d, r)
    for i in range(1, 4):
        s.dcb(f"recon_generation_net.conv.{i}", r, r)
    s.conv("recon_generation_net.head", r, CH_SRC)
    return s


def dmci_spec():
    s = Spec()
    c, n, z = DMCI_CH, DMCI_N, DMCI_CH_Z
    s.add("q_scale_enc", (QP_NUM, c, 1, 1), "q")
    s.add("q_scale_dec", (QP_NUM, c, 1, 1), "q")
    s.bit_estimator("bit_estimator_z", QP_NUM, z)
    s.dcb("enc.enc_1", CH_SRC, c)
    for i in range(6):
        s.dcb(f"enc.enc_2.{i}", c, c)
    s.conv("enc.enc_2.6", c, n, k=3)
    s.dcb("hyper_enc.0", n, z)
    s.res_down("hyper_enc.1", z, z)
    s.res_down("hyper_enc.2", z, z)
    s.res_up("hyper_dec.0", z, z)
    s.res_up("hyper_dec.1", z, z)
    s.dcb("hyper_dec.2", z, n)
    s.dcb("y_prior_fusion.0", n, 2 * n)
    s.dcb("y_prior_fusion.1", 2 * n, 2 * n)
    s.dcb("y_prior_fusion.2", 2 * n, 2 * n)
    s.conv("y_prior_fusion.3", 2 * n, 2 * n + 2)
    s.conv("y_spatial_prior_reduction", 2 * n + 2, n)
    for i in (1, 2, 3):
        s.dcb(f"y_spatial_prior_adaptor_{i}", 2 * n, 2 * n, force_adaptor=True)
    for i in range(3):
        s.dcb(f"y_spatial_prior.{i}", 2 * n, 2 * n)
    s.conv("y_spatial_prior.3", 2 * n, 2 * n)
    s.res_up("dec.dec_1.0", n, c)
    for i in range(1, 13):
        s.dcb(f"dec.dec_1.{i}", c, c)
    s.dcb("dec.dec_2", c, CH_SRC)
    return s


def spec_for(model):
    return {"dmc": dmc_spec, "dmci": dmci_spec}[model]()
