"""VAE decoder.  CPU: the oracle (oracle/vae_oracle.py) against goldens captured from the REAL reference
`AutoencoderKL` (tools/make_golden.py vae_cases); state-dict names/shapes of the drop-in class.  GPU: the HIP decode
through the drop-in `AutoencoderKL.decode` / `DenoiseModel.decode_first_stage_2DAE` boundary against the same goldens
and against the oracle on fresh inputs.  Tolerances: oracle fp32 vs reference fp32 2e-5; HIP fp16-storage 3e-3 * max|ref| (1.5 x the largest observed, 1.8e-3: gpurun_out/r3_errlog.txt)."""
import numpy as np
import pytest
import torch

from helpers import golden, inp, relerr, state_dict_for
from oracle import vae_oracle as VO

VAE_DD = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
CASES = [("vae_reduced", 64), ("vae_full_small", 128)]
TOL_HIP = 3e-3


def _model(ch):
    from moca_video_amd.vae import AutoencoderKL
    return AutoencoderKL(ddconfig=dict(VAE_DD, ch=ch), lossconfig={"target": "torch.nn.Identity"}, embed_dim=4)


def test_state_dict_names_match_reference_layout():
    m = _model(128)
    sd = m.state_dict()
    # spot checks of the reference's names/shapes (ae_modules.py:364-531, autoencoder.py:34-35)
    assert sd["decoder.conv_in.weight"].shape == (512, 4, 3, 3)
    assert sd["decoder.mid.attn_1.q.weight"].shape == (512, 512, 1, 1)
    assert sd["decoder.up.1.block.0.nin_shortcut.weight"].shape == (256, 512, 1, 1)
    assert sd["decoder.up.3.upsample.conv.weight"].shape == (512, 512, 3, 3)
    assert "decoder.up.0.upsample.conv.weight" not in sd
    assert sd["decoder.conv_out.weight"].shape == (3, 128, 3, 3)
    assert sd["encoder.down.2.downsample.conv.weight"].shape == (512, 512, 3, 3)
    assert sd["encoder.conv_out.weight"].shape == (8, 512, 3, 3)
    assert sd["quant_conv.weight"].shape == (8, 8, 1, 1) and sd["post_quant_conv.weight"].shape == (4, 4, 1, 1)
    n = sum(v.numel() for v in sd.values())
    assert n == 83_653_863, n          # AutoencoderKL of inference_t2v_512_v2.0.yaml


@pytest.mark.parametrize("name,ch", CASES)
def test_oracle_vs_reference_golden(name, ch):
    g = golden(name)
    sd = state_dict_for(_model(ch), 5)
    z = inp(name + ":z", tuple(int(v) for v in g["z_shape"]), seed=5)
    out = VO.decode_first_stage_2DAE(sd, z, float(g["scale_factor"]))
    assert out.shape == g["out"].shape
    assert relerr(out, g["out"]) < 2e-5


@pytest.mark.parametrize("name,ch", CASES)
def test_oracle_encode_vs_reference_golden(name, ch):
    """AutoencoderKL.encode of the real reference: distribution parameters, mode, and scale_factor * sample(fixed noise)"""
    g = golden(name)
    sd = state_dict_for(_model(ch), 5)
    shape = tuple(int(v) for v in g["z_shape"])
    img, nz = _enc_inputs(name, shape)
    mom = torch.cat([VO.encode_moments(sd, img[:, :, i]).unsqueeze(2) for i in range(img.shape[2])], 2)
    assert relerr(mom, g["enc_moments"]) < 2e-5
    assert relerr(torch.chunk(mom, 2, dim=1)[0], g["enc_mode"]) < 2e-5
    assert relerr(VO.encode_first_stage_2DAE(sd, img, float(g["scale_factor"]), nz), g["enc_sample"]) < 2e-5


def _enc_inputs(name, shape):
    xs = (1, 3, shape[2], 8 * shape[3], 8 * shape[4])
    return (inp(name + ":img", xs, seed=5) * 0.5).clamp(-1, 1), inp(name + ":enc_noise", shape, seed=5)


@pytest.mark.gpu
@pytest.mark.parametrize("name,ch", CASES)
def test_hip_encode_vs_reference_golden(name, ch):
    """HIP encoder (incl. the asymmetric-padding stride-2 Downsample convs) through AutoencoderKL.encode and
    DenoiseModel.encode_first_stage_2DAE against the real reference's moments / mode / scaled sample"""
    from moca_video_amd import DenoiseModel
    from helpers import REDUCED
    g = golden(name)
    shape = tuple(int(v) for v in g["z_shape"])
    img, nz = _enc_inputs(name, shape)
    dm = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": REDUCED},
                      first_stage_config={"target": "lvdm.models.autoencoder.AutoencoderKL",
                                          "params": {"embed_dim": 4, "ddconfig": dict(VAE_DD, ch=ch), "lossconfig": {"target": "torch.nn.Identity"}}},
                      scale_factor=float(g["scale_factor"]))
    dm.first_stage_model.load_state_dict(state_dict_for(dm.first_stage_model, 5), strict=True)
    dm = dm.cuda()
    frames = img[0].permute(1, 0, 2, 3).contiguous().cuda()                      # [t, 3, H, W]
    for it in range(3):                                                          # eager, capture, replay
        post = dm.first_stage_model.encode(frames)
        mom = post.parameters.cpu().permute(1, 0, 2, 3).unsqueeze(0)             # -> [1, 8, t, h, w]
        e = relerr(mom, g["enc_moments"])
        assert e < TOL_HIP, f"{name} pass {it}: moments rel err {e:.3e}"
    assert relerr(post.mode().cpu().permute(1, 0, 2, 3).unsqueeze(0), g["enc_mode"]) < TOL_HIP
    z = dm.encode_first_stage_2DAE(img.cuda(), noise=nz.cuda())
    assert z.shape == shape and relerr(z.cpu(), g["enc_sample"]) < TOL_HIP
    assert relerr(post.std.cpu(), torch.exp(0.5 * torch.clamp(torch.chunk(post.parameters.cpu(), 2, 1)[1], -30, 20))) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name,ch", CASES)
def test_hip_decode_vs_reference_golden(name, ch):
    g = golden(name)
    m = _model(ch)
    m.load_state_dict(state_dict_for(m, 5), strict=True)
    m = m.cuda()
    z = inp(name + ":z", tuple(int(v) for v in g["z_shape"]), seed=5).cuda()
    zz = (1.0 / float(g["scale_factor"]) * z)[0].permute(1, 0, 2, 3).contiguous()      # [t, c, h, w]
    ref = torch.from_numpy(g["out"])[0].permute(1, 0, 2, 3)
    for it in range(3):                                  # eager, graph capture, graph replay
        out = m.decode(zz)
        assert out.shape == ref.shape
        e = relerr(out.cpu(), ref)
        assert e < TOL_HIP, f"{name} pass {it}: rel err {e:.3e}"
    assert any(p.graph is not None for p in m._plans.values()), "hipGraph replay path was not taken"


@pytest.mark.gpu
def test_hip_decode_first_stage_vs_oracle_fresh_inputs():
    """full-width decoder, 2 frames of 16x24 latents that no golden holds, through DenoiseModel.decode_first_stage_2DAE"""
    from moca_video_amd import DenoiseModel
    from helpers import REDUCED
    dm = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": REDUCED},
                      first_stage_config={"target": "lvdm.models.autoencoder.AutoencoderKL",
                                          "params": {"embed_dim": 4, "ddconfig": VAE_DD, "lossconfig": {"target": "torch.nn.Identity"}}},
                      scale_factor=0.18215)
    sd = state_dict_for(dm.first_stage_model, 9)
    dm.first_stage_model.load_state_dict(sd, strict=True)
    dm = dm.cuda()
    z = inp("vae.fresh.z", (1, 4, 2, 16, 24), seed=9)
    ref = VO.decode_first_stage_2DAE(sd, z, 0.18215)
    out = dm.decode_first_stage_2DAE(z.cuda())
    assert out.shape == ref.shape == (1, 3, 2, 128, 192)
    assert relerr(out.cpu(), ref) < TOL_HIP


@pytest.mark.gpu
def test_decode_rejects_bad_inputs():
    m = _model(64)
    m.load_state_dict(state_dict_for(m, 5), strict=True)
    m = m.cuda()
    with pytest.raises(ValueError):
        m.decode(torch.zeros(1, 3, 8, 8, device="cuda"))
    with pytest.raises(ValueError):
        m.decode(torch.zeros(1, 4, 8, 8))
    with pytest.raises(NotImplementedError):
        m.decode(torch.zeros(1, 4, 5, 8, device="cuda"))
    with pytest.raises(ValueError):
        m.encode(torch.zeros(1, 4, 64, 64, device="cuda"))
    with pytest.raises(NotImplementedError):
        m.encode(torch.zeros(1, 3, 60, 64, device="cuda"))


@pytest.mark.gpu
def test_prompt_mode_driver_end_to_end(tmp_path):
    """moca_video_amd.io.run_prompts = the prompt-mode loop of videocrafter_main.py:176-232 on the drop-in classes:
    CSV -> rank striding -> base DDIM sampling (writes the {0,N}.pt latent cache) -> MoCA FIFO with mask injection ->
    VAE decode of the emitted frames -> GIF; a second run must find the latent cache and skip the base sampling."""
    import os
    import types
    from PIL import Image
    from moca_video_amd import DenoiseModel
    from moca_video_amd.io import run_prompts
    from helpers import REDUCED
    dm = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": REDUCED},
                      first_stage_config={"target": "lvdm.models.autoencoder.AutoencoderKL",
                                          "params": {"embed_dim": 4, "ddconfig": dict(VAE_DD, ch=64), "lossconfig": {"target": "torch.nn.Identity"}}},
                      scale_factor=0.18215)
    dm.model.diffusion_model.load_state_dict(state_dict_for(dm.model.diffusion_model, 11), strict=True)
    dm.first_stage_model.load_state_dict(state_dict_for(dm.first_stage_model, 5), strict=True)
    dm = dm.cuda()
    csv = tmp_path / "p.csv"
    csv.write_text('prompt,conditioned_object,conditioned_image_path,conditioned_prompt,gamma\n'
                   '"a cat, sitting",cat,assets/a.jpg,"the condition is a dog",2\n'
                   'second prompt,obj,assets/b.jpg,"the condition is a bird",1.5\n'
                   'third prompt,obj,assets/c.jpg,"the condition is a fish",1.5\n')
    args = types.SimpleNamespace(prompt_file=str(csv), prompt_index=None, rank=1, num_processes=2, height=64, width=64, fps=10,
                                 video_length=16, num_partitions=4, num_inference_steps=64, new_video_length=8,
                                 lookahead_denoising=True, eta=1.0, unconditional_guidance_scale=12.0, output_dir=None,
                                 use_self_attention=False, output_fps=10)
    embed = lambda text: inp("txt:" + text, (1, 77, 128)).cuda()
    cimg = lambda row: inp("cimg:" + row["conditioned_object"], (1, 4, 1, 8, 8)).cuda()
    mask = lambda row, shape: (inp("mask:" + row["prompt"], shape) > 0.3).float().cuda()
    done = run_prompts(args, dm, embed, cimg, mask, root=str(tmp_path), uc_emb=embed(""), n_iterations=6)
    assert list(done) == [1]                                       # rank 1 of 2 owns row 1 only
    im = Image.open(done[1])
    assert im.n_frames == 4 and im.size == (64, 64)                # last new_video_length//2 emitted frames, 8x latents
    lat = os.path.join(str(tmp_path), "results/videocraft_v2_fifo/latents/64steps/second prompt/eta1.0")
    assert os.path.exists(lat + "/0.pt") and os.path.exists(lat + "/64.pt")
    assert os.path.exists(os.path.dirname(done[1]) + "/origin.gif")
    os.remove(os.path.dirname(done[1]) + "/origin.gif")
    done2 = run_prompts(args, dm, embed, cimg, mask, root=str(tmp_path), uc_emb=embed(""), n_iterations=2)
    assert not os.path.exists(os.path.dirname(done2[1]) + "/origin.gif")      # latent cache hit: no base sampling


@pytest.mark.gpu
def test_prepare_latents_davis_branch_encodes_frames():
    """funcs.py:38-48: DAVIS mode builds the FIFO queue from the VAE encoding of the video frames (RGBA -> RGB)."""
    import types
    from moca_video_amd import DenoiseModel
    from moca_video_amd.fifo import prepare_latents
    from moca_video_amd.sampler import DDIMSampler
    from helpers import REDUCED
    dm = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": REDUCED},
                      first_stage_config={"target": "lvdm.models.autoencoder.AutoencoderKL",
                                          "params": {"embed_dim": 4, "ddconfig": dict(VAE_DD, ch=64), "lossconfig": {"target": "torch.nn.Identity"}}},
                      scale_factor=0.18215)
    sd = state_dict_for(dm.first_stage_model, 5)
    dm.first_stage_model.load_state_dict(sd, strict=True)
    dm = dm.cuda()
    s = DDIMSampler(dm)
    s.make_schedule(ddim_num_steps=16, ddim_eta=1.0, verbose=False)
    args = types.SimpleNamespace(num_inference_steps=16, video_length=8, lookahead_denoising=True)
    frames = (inp("davis.frames", (1, 4, 8, 64, 64)) * 0.5).clamp(-1, 1)
    noises = [torch.zeros(1, 4, 1, 8, 8) for _ in range(20)]                 # zero queue noise: queue = sqrt(a_j) * z
    torch.manual_seed(0)
    lat = prepare_latents(args, None, s, model=dm, data=(frames, None), noises=noises)
    assert lat.shape == (1, 4, 20, 8, 8)
    # posterior mean * scale (the sample noise is random): compare the direction of the first queue frame with the oracle's mode
    z_mode = VO.encode_first_stage_2DAE(sd, frames[:, :3], 0.18215)
    a0 = float(s.ddim_alphas[0])
    got = lat[:, :, 0].cpu() / a0 ** 0.5
    std = torch.exp(0.5 * torch.clamp(torch.chunk(VO.encode_moments(sd, frames[:, :3, 0]), 2, 1)[1], -30, 20)) * 0.18215
    assert ((got - z_mode[:, :, 0]).abs() <= 6 * std + 2e-2 * z_mode.abs().max()).all()


@pytest.mark.gpu
def test_davis_mode_driver_end_to_end(tmp_path):
    """moca_video_amd.io.run_davis = the DAVIS branch of videocrafter_main.py:102-175: frames + annotation masks from disk ->
    VAE-encoded queue -> MoCA FIFO with the DAVIS masks -> decoded GIF of the first new_video_length//2 frames"""
    import types
    from PIL import Image
    from moca_video_amd import DenoiseModel
    from moca_video_amd.io import run_davis
    from helpers import REDUCED
    fd = tmp_path / "DAVIS" / "JPEGImages" / "480p" / "bear"
    md = tmp_path / "DAVIS" / "Annotations" / "480p" / "bear"
    fd.mkdir(parents=True); md.mkdir(parents=True)
    rng = np.random.default_rng(0)
    for i in range(24):
        Image.fromarray(rng.integers(0, 255, (48, 80, 3), dtype=np.uint8)).save(str(fd / f"{i:05d}.jpg"))
        m = np.zeros((48, 80), np.uint8)
        m[12:30, 20 + i:50 + i] = 1
        Image.fromarray(m).save(str(md / f"{i:05d}.png"))
    dm = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": REDUCED},
                      first_stage_config={"target": "lvdm.models.autoencoder.AutoencoderKL",
                                          "params": {"embed_dim": 4, "ddconfig": dict(VAE_DD, ch=64), "lossconfig": {"target": "torch.nn.Identity"}}},
                      scale_factor=0.18215)
    dm.model.diffusion_model.load_state_dict(state_dict_for(dm.model.diffusion_model, 11), strict=True)
    dm.first_stage_model.load_state_dict(state_dict_for(dm.first_stage_model, 5), strict=True)
    dm = dm.cuda()
    args = types.SimpleNamespace(video_name="bear", davis_root=str(tmp_path / "DAVIS"), height=64, width=64, fps=10, video_length=8,
                                 num_partitions=2, num_inference_steps=16, new_video_length=6, lookahead_denoising=True, eta=1.0,
                                 unconditional_guidance_scale=12.0, output_dir=None, use_self_attention=False, output_fps=10,
                                 sampling_strategy="first", gamma=0.5, use_davis=True)
    embed = lambda text: inp("txt:" + text, (1, 77, 128)).cuda()
    cimg = inp("davis.cimg", (1, 4, 1, 8, 8)).cuda()
    path = run_davis(args, dm, embed, "a bear walking, cat.", cond_image=cimg, root=str(tmp_path), uc_emb=embed(""), n_iterations=4)
    im = Image.open(path)
    assert im.n_frames == 3 and im.size == (64, 64)           # first new_video_length // 2 of the 4 emitted frames


def _davis_shift_inputs():
    h, w, Q = 8, 8, 72
    frames = (inp("fifo.davis.frames", (1, 4, 3, 8 * h, 8 * w)) * 0.5).clamp(-1, 1)
    masks = (inp("fifo.davis.masks", (1, 1, Q, h, w)) > 0.3).float()
    lat = inp("fifo.davis.lat", (1, 4, Q, h, w))
    anchor_nz = [inp(f"fifo.davis.anchor_nz{i}", (1, 4, h, w)) for i in range(2)]
    nz = [inp(f"fifo.davis.nz{i}", (1, 4, h, w)) for i in range(2)]
    return frames, masks, lat, anchor_nz, nz


def test_oracle_davis_shift_latents_vs_reference_golden():
    """DAVIS branch of shift_latents (funcs.py:101-118) of the REAL reference, two consecutive shifts: anchor = posterior
    sample of the VAE encoding of the last DAVIS frame, FreeInit mix into the queue tail, mask queue shift"""
    from oracle import freeinit_oracle as FO
    g = golden("fifo_davis_shift")
    sd = state_dict_for(_model(64), 5)
    frames, masks, lat, anchor_nz, nz = _davis_shift_inputs()
    for i in range(2):
        enc = lambda x, _i=i: VO.encode_first_stage_2DAE(sd, x, 0.18215, anchor_nz[_i].unsqueeze(2))
        lat, masks = FO.shift_latents(lat, nz[i], davis_data=(frames, masks), encode=enc)
        assert relerr(lat, g[f"lat{i + 1}"]) < 2e-5
        assert torch.equal(masks, torch.from_numpy(g[f"masks{i + 1}"]))


@pytest.mark.gpu
def test_hip_davis_shift_latents_vs_reference_golden():
    """moca_video_amd.fifo.shift_latents(davis_data=...) (HIP VAE encoder + FreeInit kernels) against the same golden"""
    from moca_video_amd import DenoiseModel
    from moca_video_amd.fifo import shift_latents
    from helpers import REDUCED
    g = golden("fifo_davis_shift")
    dm = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": REDUCED},
                      first_stage_config={"target": "lvdm.models.autoencoder.AutoencoderKL",
                                          "params": {"embed_dim": 4, "ddconfig": dict(VAE_DD, ch=64), "lossconfig": {"target": "torch.nn.Identity"}}},
                      scale_factor=0.18215)
    dm.first_stage_model.load_state_dict(state_dict_for(dm.first_stage_model, 5), strict=True)
    dm = dm.cuda()
    frames, masks, lat, anchor_nz, nz = _davis_shift_inputs()
    lat, data = lat.cuda(), (frames.cuda(), masks.cuda())
    for i in range(2):
        lat, data = shift_latents(lat, data, dm, noise=nz[i].cuda(), anchor_noise=anchor_nz[i].unsqueeze(2).cuda())
        assert relerr(lat.cpu(), g[f"lat{i + 1}"]) < TOL_HIP
        assert torch.equal(data[1].cpu(), torch.from_numpy(g[f"masks{i + 1}"]))
