"""CPU: the prompt CSV / rank striding / directory / GIF helpers of moca_video_amd.io against the reference's format
(`prompts/prompts.csv` header and quoting, funcs.py:506-535; videocrafter_main.py:25-55,179-181)."""
import types

import numpy as np
import pytest
import torch

CSV = '''prompt,conditioned_object,conditioned_image_path,conditioned_prompt,gamma
"An astronaut floating in space, wearing a detailed white spacesuit, Earth visible",astronaut,"assets/robot.jpg","the condition is a robot", 2
"A superhero in a dynamic pose against a city skyline, cape flowing in the wind",superhero,"assets/eagle.jpg","the condition is an eagle",1.5
plain prompt,obj , assets/x.jpg ,,0.5
'''


def _csv(tmp_path):
    p = tmp_path / "prompts.csv"
    p.write_text(CSV)
    return str(p)


def test_load_prompts_format(tmp_path):
    from moca_video_amd.io import load_prompts
    rows = load_prompts(_csv(tmp_path))
    assert len(rows) == 3
    assert rows[0]["prompt"].startswith("An astronaut floating in space, wearing")          # quoted commas survive
    assert rows[0] == {"prompt": rows[0]["prompt"], "conditioned_object": "astronaut", "conditioned_image_path": "assets/robot.jpg",
                       "conditioned_prompt": "the condition is a robot.", "gamma": 2.0}
    assert rows[1]["gamma"] == 1.5 and rows[2]["conditioned_object"] == "obj" and rows[2]["conditioned_prompt"] == "."
    assert load_prompts(_csv(tmp_path), prompt_index=1) == [rows[1]]
    with pytest.raises(ValueError):
        load_prompts(_csv(tmp_path), prompt_index=3)


def test_rank_striding_partitions_every_prompt_once():
    from moca_video_amd.io import shard_indices
    for n, world in ((64, 8), (10, 4), (3, 8)):
        parts = [shard_indices(n, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert parts[1] == list(range(n))[1::world]
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_set_directory_convention(tmp_path):
    from moca_video_amd.io import set_directory
    a = types.SimpleNamespace(output_dir=None, use_self_attention=False, eta=1.0, new_video_length=100, lookahead_denoising=True,
                              num_partitions=4, video_length=16, num_inference_steps=64)
    out, lat = set_directory(a, "a cat", root=str(tmp_path))
    assert out.endswith("results/videocraft_v2_fifo/random_noise/sam2/a cat")
    assert lat.endswith("results/videocraft_v2_fifo/latents/64steps/a cat/eta1.0")
    a.eta, a.new_video_length, a.num_partitions = 0.5, 60, 8
    out, _ = set_directory(a, "a cat", root=str(tmp_path))
    assert out.endswith("sam2/a cat/n=8/eta0.5/60frames")


def test_gif_and_frame_writers(tmp_path):
    from PIL import Image
    from moca_video_amd.fifo import tensor2image
    from moca_video_amd.io import frames_to_uint8, save_frames, save_gif
    v = torch.linspace(-1.5, 1.5, 3 * 5 * 8 * 12).reshape(1, 3, 5, 8, 12)
    u8 = frames_to_uint8(v)
    assert u8.shape == (5, 8, 12, 3) and u8.dtype == np.uint8 and u8.min() == 0 and u8.max() == 255
    path = save_gif(v, str(tmp_path), "clip", duration_ms=100)
    im = Image.open(path)
    assert im.n_frames == 5 and im.size == (12, 8)
    img = tensor2image(v[:, :, [2]])
    assert np.array_equal(np.asarray(img), u8[2])
    save_frames([img, u8[0]], str(tmp_path / "frames"))
    assert Image.open(str(tmp_path / "frames" / "1.png")).size == (12, 8)


def test_load_cond_image_is_rgba_resized_to_latent_grid(tmp_path):
    from PIL import Image
    from moca_video_amd.io import load_cond_image
    rgb = (np.arange(96 * 128 * 3) % 251).astype(np.uint8).reshape(96, 128, 3)
    Image.fromarray(rgb).save(str(tmp_path / "c.png"))
    t = load_cond_image(str(tmp_path / "c.png"), 320, 512, device="cpu")
    assert t.shape == (1, 4, 1, 40, 64) and t.dtype == torch.float32
    assert float(t.min()) >= 0.0 and float(t.max()) <= 1.0
    assert torch.all(t[0, 3] == 1.0)                         # opaque alpha from convert("RGBA")
    ref = np.asarray(Image.fromarray(rgb).convert("RGBA").resize((64, 40), Image.BILINEAR), dtype=np.float32) / 255.0
    assert np.allclose(t[0, :, 0].permute(1, 2, 0).numpy(), ref)


def test_load_masks_from_disk(tmp_path):
    from PIL import Image
    from moca_video_amd.io import load_masks
    m = np.zeros((64, 128), np.uint8)
    m[16:48, 32:96] = 255
    Image.fromarray(m).save(str(tmp_path / "0.png"))
    np.save(str(tmp_path / "2.npy"), np.stack([m > 0, np.zeros_like(m) > 0]).astype(np.float32))
    out = load_masks(str(tmp_path), 4, 64, 128, device="cpu")
    assert out[1] is None and out[3] is None
    assert out[0].shape == (1, 8, 16) and out[2].shape == (2, 8, 16)
    assert out[0].sum() == 4 * 8 and torch.equal(out[0][0], out[2][0]) and out[2][1].sum() == 0


def test_load_davis_data_layout(tmp_path):
    from PIL import Image
    from moca_video_amd.io import load_davis_data
    fd = tmp_path / "JPEGImages" / "480p" / "bear"
    md = tmp_path / "Annotations" / "480p" / "bear"
    fd.mkdir(parents=True); md.mkdir(parents=True)
    for i in range(20):
        Image.fromarray(np.full((48, 80, 3), 10 * i, np.uint8)).save(str(fd / f"{i:05d}.jpg"))
        m = np.zeros((48, 80), np.uint8)
        m[10:30, 20:50] = i % 3          # palette index: frames with i % 3 == 0 have an empty mask
        Image.fromarray(m).save(str(md / f"{i:05d}.png"))
    frames, masks = load_davis_data("bear", str(tmp_path), video_size=(4, 8), video_frames=16)
    assert frames.shape == (1, 4, 16, 32, 64) and masks.shape == (1, 1, 16, 4, 8)
    assert float(frames.min()) >= -1.0 and float(frames.max()) <= 1.0 and torch.all(frames[0, 3] == 1.0)
    assert set(masks.unique().tolist()) <= {0.0, 1.0} and masks[0, 0, 0].sum() == 0 and masks[0, 0, 1].sum() > 0
    f2, _ = load_davis_data("bear", str(tmp_path), video_size=(4, 8), video_frames=8, sampling_strategy="uniform")
    assert f2.shape[2] == 8 and abs(float(f2[0, 0, 1].mean()) - ((20 / 255 - 0.5) * 2)) < 0.05    # stride 2 -> frame 2
    with pytest.raises(ValueError):
        load_davis_data("bear", str(tmp_path), sampling_strategy="nope")


def test_uncond_embedding_follows_uncond_type():
    """funcs.py:199-206: `uncond_type == "empty_seq"` (the YAML) is the text encoding of "" -- the caller must supply it; only
    "zero_embed" is zeros.  A missing uc_emb must not silently become zeros."""
    from moca_video_amd.fifo import uncond_embedding
    c = torch.randn(1, 77, 16)
    uc = torch.randn(1, 77, 16)
    assert uncond_embedding(types.SimpleNamespace(uncond_type="empty_seq"), c, uc) is uc
    with pytest.raises(ValueError):
        uncond_embedding(types.SimpleNamespace(uncond_type="empty_seq"), c, None)
    z = uncond_embedding(types.SimpleNamespace(uncond_type="zero_embed"), c, None)
    assert z.shape == c.shape and not z.any()
    with pytest.raises(NotImplementedError):
        uncond_embedding(types.SimpleNamespace(uncond_type="other"), c, None)


def test_driver_default_unconditional_is_the_empty_prompt():
    from moca_video_amd.io import _empty_prompt_embedding
    calls = []
    embed = lambda s: calls.append(s) or torch.full((1, 77, 4), float(len(s)))
    m = types.SimpleNamespace(uncond_type="empty_seq")
    uc = _empty_prompt_embedding(m, embed, None)
    assert calls == [""] and float(uc.max()) == 0.0
    given = torch.ones(1, 77, 4)
    assert _empty_prompt_embedding(m, embed, given) is given and calls == [""]
    assert _empty_prompt_embedding(types.SimpleNamespace(uncond_type="zero_embed"), embed, None) is None


def test_kept_frames_slice_is_the_reference_expression():
    """videocrafter_main.py:228-230 keeps `video_frames[-args.new_video_length//2:]`: for odd lengths -N//2 floors (-101//2 == -51)"""
    import inspect
    from moca_video_amd import io
    assert "frames[-args.new_video_length // 2:]" in inspect.getsource(io.run_prompts)
    frames = list(range(148))
    assert len(frames[-101 // 2:]) == 51 and len(frames[-100 // 2:]) == 50
