"""GPT-2 img2txt path (config C5): oracle pinned against the reference's own GPT2LMHeadModel +
sample_sequence (build container), engine decode vs oracle on the GPU (token-exact up to fp32 near-ties)."""
import numpy as np
import pytest
import torch

from clip_glass_amd import synth
from oracle import gpt2_ref
from util import diag

MINI = dict(n_embd=128, n_layer=2, vocab=2048)


def _t(sd):
    return {k: torch.as_tensor(v) for k, v in sd.items()}


def _ctx(seed, P, n, vocab):
    return np.random.RandomState(seed).randint(0, vocab, size=(P, n)).astype(np.int64)


@pytest.mark.reference
def test_oracle_gpt2_matches_reference_sampling():
    import ref_harness as rh
    if not rh.available():
        pytest.skip("/root/reference not present")
    sd = synth.make_state(synth.gpt2_spec(**MINI, n_positions=64), 2)
    model, sample_sequence = rh.build_ref_gpt2(sd, MINI["n_embd"], MINI["n_layer"], MINI["vocab"])
    ctx = torch.tensor(_ctx(1, 6, 23, MINI["vocab"]))
    ref = sample_sequence(model=model, length=12, context=ctx, start_token=None, batch_size=6, temperature=0.7,
                          top_k=40, device="cpu", sample=False)
    ora = gpt2_ref.sample_sequence(_t(sd), ctx, 12)
    assert np.array_equal(np.asarray(ref), ora.numpy())
    with torch.no_grad():
        lr, _ = model(ctx)
        lo, _ = gpt2_ref.forward(_t(sd), ctx)
    np.testing.assert_allclose(lo.numpy(), lr.numpy(), rtol=2e-4, atol=2e-5)


def test_gpt2_bpe_known_answers():
    import os
    enc, voc = "/root/reference/gpt2/weights/encoder.json", "/root/reference/gpt2/weights/vocab.bpe"
    if not os.path.exists(enc):
        pytest.skip("GPT-2 BPE assets not present")
    from clip_glass_amd.gpt2_bpe import Gpt2Bpe
    b = Gpt2Bpe(enc, voc)
    assert b.encode("the picture of") == [1169, 4286, 286]
    for s in ["the picture of a dog, really!", "Hello  world\n\nnew — line", "naïve café 123"]:
        assert b.decode(b.encode(s)) == s
    try:
        import ref_harness as rh
        import importlib, types
        rh.load_reference()
        cwd = os.getcwd(); os.chdir(rh.REF)
        try:
            ge = importlib.import_module("gpt2.encoder")
        finally:
            os.chdir(cwd)
        ref = ge.get_encoder(types.SimpleNamespace(encoder=enc, vocab=voc))
        for s in ["a wolf at night with the moon", "It's 42°C — isn't it?", "  leading spaces"]:
            assert b.encode(s) == ref.encode(s)
    except ImportError:
        pass


def test_generated_bpe_assets_have_the_real_sizes_and_round_trip(tmp_path):
    """synth.write_bpe_assets: the vocabularies `bench.py --config gpt2` runs the host stage on where the reference's data files are
    absent — 50257 GPT-2 ids (<|endoftext|> = 50256, config.py:28) and 49408 CLIP ids in the reference's file formats; the prompt is
    3 tokens (models.py:30); text -> ids -> text round-trips; a decoded 30-token sample fits clip.tokenize's 77-token context."""
    from clip_glass_amd.gpt2_bpe import Gpt2Bpe
    from clip_glass_amd.tokenizer import ClipTokenizer
    ej, vb, cb = synth.write_bpe_assets(str(tmp_path))
    g, c = Gpt2Bpe(ej, vb), ClipTokenizer(cb)
    assert len(g.token_id) == 50257 and g.eot == 50256 and len(c.token_id) == 49408 and c.eot == 49407 and c.sot == 49406
    assert len(g.encode("the picture of")) == 3
    for s in ["the picture of", "a wolf at night with the moon", "hello world"]:
        assert g.decode(g.encode(s)) == s
    toks = np.random.RandomState(0).randint(0, 50256, (16, 30))
    texts = [g.decode(r)[:50] for r in toks]
    tk = c.tokenize(texts)
    assert tk.shape == (16, 77) and (tk[:, 0] == c.sot).all() and ((tk == c.eot).sum(1) == 1).all()


def test_parse_out_semantics():
    """models.py:32-42: latent tokens dropped, cut at <|endoftext|>, empty text when EOT sits in the latent part."""
    dec = lambda toks: " ".join(str(t) for t in toks)
    out = [[5, 6, 7, 8, 9, 99, 10], [99, 6, 7, 8, 9, 10, 11], [5, 6, 7, 8, 9, 10, 11]]
    texts = gpt2_ref.parse_out(out, 3, 99, dec, 50)
    assert texts == ["8 9", "", "8 9 10 11"]
    assert gpt2_ref.parse_out([[1, 2, 3] + list(range(100, 140))], 3, 99, dec, 10)[0] == "100 101 10"


def _assert_token_parity(tag, got, ora, margins, n_ctx):
    """Greedy decode is an INDEX result: the bar is token-for-token equality with the oracle on every row.  The one
    escape is a near-tie: a row may leave the oracle's sequence only at a step where the oracle's own top-2 logit margin is below
    1e-4 (fp32 summation order decides such a pick); every such row is logged with its step and margin.  Returns the excepted rows."""
    assert got.shape == ora.shape
    excepted = []
    for p in range(got.shape[0]):
        same = got[p] == ora[p]
        if same.all():
            continue
        first = int(np.argmin(same)) - n_ctx
        assert first >= 0, "%s: row %d differs inside its context" % (tag, p)
        m = float(margins[p, first])
        diag("[gpt2] %s NEAR-TIE EXCEPTION: row %d leaves the oracle at step %d, oracle top-2 margin %.3e" % (tag, p, first, m))
        assert m < 1e-4, "%s: row %d diverged at step %d with a clear margin %.3e" % (tag, p, first, m)
        excepted.append(p)
    diag("[gpt2] %s: %d/%d rows token-identical to the oracle, %d near-tie exception(s); min top-2 margin over all steps %.3e"
         % (tag, got.shape[0] - len(excepted), got.shape[0], len(excepted), float(np.min(margins))))
    return excepted


@pytest.mark.gpu
@pytest.mark.parametrize("geo,P,n_ctx,length", [(MINI, 8, 23, 10), (dict(n_embd=256, n_layer=3, vocab=5000), 16, 23, 30)])
def test_engine_gpt2_decode_matches_oracle(geo, P, n_ctx, length):
    import glass_models as M
    from clip_glass_amd.engine import Engine
    sd = synth.make_state(synth.gpt2_spec(**geo, n_positions=64), 2)
    clip = M.CONFIGS["mini"]["clip"]
    sd.update(synth.make_state(synth.clip_visual_spec(clip[0], clip[1], clip[3], clip[4], clip[5]), 0))
    e = Engine([], latent_size=4, mapping_layers=0, batch_size=1, use_discriminator=False, n_obj=1, max_pop=P, clip=clip,
               noise_mode=0)
    e.load_state(sd)
    e.finalize()
    ctx = _ctx(1, P, n_ctx, geo["vocab"])
    got = e.gpt2_decode(ctx, length)
    e.close()
    detail = {}
    ora = gpt2_ref.sample_sequence(_t(sd), torch.tensor(ctx), length, detail=detail).numpy()
    assert got.shape == ora.shape and np.array_equal(got[:, :n_ctx], ctx)
    _assert_token_parity("decode P=%d L=%d" % (P, length), got, ora, detail["margins"].numpy() if hasattr(detail["margins"], "numpy") else detail["margins"], n_ctx)


@pytest.mark.gpu
def test_engine_gpt2_decode_rows_do_not_depend_on_the_launch():
    """A sequence's tokens are a function of its own context only: P = 72 in one call (decoded as row groups of 64 + 8) equals the
    same rows decoded as shards of 8 / 64 / 72 rows in any grouping — prefill products never take the M <= 64 split-K path, the
    single-token step products pick their split from (N, K) alone."""
    import glass_models as M
    from clip_glass_amd.engine import Engine
    geo = dict(n_embd=256, n_layer=3, vocab=5000)
    sd = synth.make_state(synth.gpt2_spec(**geo, n_positions=64), 2)
    clip = M.CONFIGS["mini"]["clip"]
    sd.update(synth.make_state(synth.clip_visual_spec(clip[0], clip[1], clip[3], clip[4], clip[5]), 0))
    e = Engine([], latent_size=4, mapping_layers=0, batch_size=1, use_discriminator=False, n_obj=1, max_pop=72, clip=clip, noise_mode=0)
    e.load_state(sd)
    e.finalize()
    ctx = _ctx(4, 72, 23, geo["vocab"])
    whole = e.gpt2_decode(ctx, 12)
    assert whole.shape == (72, 35)
    for lo, hi in ((0, 8), (8, 72), (0, 64), (64, 72), (3, 5)):
        np.testing.assert_array_equal(e.gpt2_decode(ctx[lo:hi], 12), whole[lo:hi])
    e.close()


def _synthetic_vocabs(tmp):
    """Tiny BPE assets in the reference's file formats (gpt2/encoder.py:107-115, clip/simple_tokenizer.py:66-72) so the
    img2txt path runs where the reference's data files are absent (the GPU box)."""
    import gzip, json, os
    from clip_glass_amd.tokenizer import _byte_alphabet
    b2c, order = _byte_alphabet()
    chars = [b2c[b] for b in order]
    merges = [("t", "h"), ("th", "e"), ("a", "n"), ("i", "n"), ("\u0120", "the"), ("\u0120", "a"), ("e", "r"), ("o", "n")]
    enc = {c: i for i, c in enumerate(chars)}
    for a, b in merges:
        enc[a + b] = len(enc)
    enc["<|endoftext|>"] = len(enc)
    ej, vb, cb = os.path.join(tmp, "encoder.json"), os.path.join(tmp, "vocab.bpe"), os.path.join(tmp, "clip_bpe.txt.gz")
    json.dump(enc, open(ej, "w"))
    open(vb, "w", encoding="utf-8").write("#version: 0.2\n" + "\n".join(a + " " + b for a, b in merges) + "\n")
    cm = [("t", "h"), ("th", "e</w>"), ("a", "n"), ("i", "n</w>"), ("o", "n</w>"), ("e", "r</w>")]
    gzip.open(cb, "wt", encoding="utf-8").write("#version\n" + "\n".join(a + " " + b for a, b in cm) + "\n")
    return ej, vb, cb, len(enc), 512 + len(cm) + 2


@pytest.mark.gpu
@pytest.mark.parametrize("assets", ["synthetic", "reference"])
def test_img2txt_generation_problem_end_to_end(assets, tmp_path):
    """GPT2 config through GenerationProblem: decode -> parse -> CLIP tokenize -> text tower -> cosine vs image feature."""
    import os, types
    if assets == "reference":
        enc, voc = "/root/reference/gpt2/weights/encoder.json", "/root/reference/gpt2/weights/vocab.bpe"
        bpe = "/root/reference/assets/bpe_simple_vocab_16e6.txt.gz"
        if not (os.path.exists(enc) and os.path.exists(bpe)):
            pytest.skip("BPE assets (reference data files) not present on this box")
        gvocab, cvocab = 50257, 49408
    else:
        enc, voc, bpe, gvocab, cvocab = _synthetic_vocabs(str(tmp_path))
    import glass_models as M
    from clip_glass_amd import config as gconfig
    from clip_glass_amd import generator as gen_mod
    from clip_glass_amd.problem import GenerationProblem
    from oracle import clip_ref
    clipg = M.CONFIGS["mini"]["clip"]
    cfg = types.SimpleNamespace(config="GPT2", device="cuda", target="unused")
    vars(cfg).update(gconfig.get_config("GPT2"))
    tf = synth.normal(3, "imgfeat", (clipg[5],))
    vars(cfg).update(weights="synthetic:2", clip_weights="synthetic:0", clip_geometry=clipg,
                     clip_text_geometry=dict(width=64, layers=2, vocab=cvocab), encoder_size=gvocab,
                     gpt2_geometry=dict(n_embd=128, n_layer=2), encoder=enc, vocab=voc, bpe_path=bpe, target_features=tf,
                     pop_size=8, max_pop=8)
    if assets == "synthetic":
        cfg.init_text = "the an"
    prob = GenerationProblem(cfg)
    x = np.random.RandomState(0).randint(0, gvocab, size=(8, 20))
    out = {}
    prob._evaluate(x, out)
    assert out["F"].shape == (8,) and out["G"].shape == (8,)
    texts = prob.generator.last_texts
    assert len(texts) == 8 and all(isinstance(t, str) and len(t) <= 50 for t in texts)
    diag("[gpt2] img2txt (%s vocab) sample texts: %r" % (assets, texts[:2]))
    # oracle: same texts -> tokenize -> oracle text tower -> cosine
    sd = synth.make_state(synth.clip_text_spec(width=64, layers=2, vocab=cvocab, out_dim=clipg[5]), 0)
    try:
        tok = prob.generator.tokenizer.tokenize(texts)
        tfeat = clip_ref.encode_text(_t(sd), torch.tensor(tok))
        sim = torch.cosine_similarity(tfeat, torch.tensor(tf)[None]).numpy()
    except Exception:
        sim = np.zeros(8, np.float32)                      # generator.py:53-56: tokenisation failure -> zeros
    np.testing.assert_allclose(out["F"], -sim, rtol=0, atol=2e-3)
    # decode parity of the same population against the oracle (tokens, then texts through parse_out)
    sdg = synth.make_state(synth.gpt2_spec(128, 2, gvocab), 2)
    ctx = np.concatenate([x, np.tile(prob.generator.model.init_tokens, (8, 1))], axis=1)
    dd = {}
    ora = gpt2_ref.sample_sequence(_t(sdg), torch.tensor(ctx), cfg.max_tokens_len, detail=dd).numpy()
    bpe_dec = prob.generator.model.enc
    ref_texts = gpt2_ref.parse_out(ora, cfg.dim_z, bpe_dec.eot, bpe_dec.decode, cfg.max_text_len)
    got_tok = prob.generator.engine.gpt2_decode(ctx, cfg.max_tokens_len)
    mg = dd["margins"].numpy() if hasattr(dd["margins"], "numpy") else dd["margins"]
    excepted = _assert_token_parity("img2txt (%s vocab)" % assets, got_tok, ora, mg, ctx.shape[1])
    assert all(a == b for i, (a, b) in enumerate(zip(texts, ref_texts)) if i not in excepted)
    prob.generator.engine.close()


@pytest.mark.gpu
def test_img2txt_full_size_generation_problem_matches_oracle(tmp_path):
    """Config C5 END TO END at the real geometry, with a checker: GPT-2-small (12 x 768, vocab 50257) token-latent
    decode -> parse_out -> clip.tokenize on generated BPE tables of the reference's sizes and file formats (synth.write_bpe_assets;
    50257 / 49408 ids) -> CLIP text tower (512 x 12 x 8 heads, context 77) -> cosine against the image feature, through
    `GenerationProblem._evaluate` at P = 64 — the chain `bench.py --config gpt2` times (/root/reference/generator.py:52-59,
    models.py:45-62, problem.py:14-29).  Oracle: gpt2_ref.sample_sequence -> gpt2_ref.parse_out -> the same tokenizer ->
    clip_ref.encode_text -> cosine; F within 1e-3 relative (north_star's bar), texts identical, tokens identical."""
    import types
    from clip_glass_amd import config as gconfig
    from clip_glass_amd.problem import GenerationProblem
    from oracle import clip_ref
    P = 64
    clipg = (768, 12, 12, 32, 224, 512)
    enc, voc, bpe = synth.write_bpe_assets(str(tmp_path))
    tf = synth.normal(3, "imgfeat", (512,))
    cfg = types.SimpleNamespace(config="GPT2", device="cuda:0", target="unused")
    vars(cfg).update(gconfig.get_config("GPT2"))
    vars(cfg).update(weights="synthetic:5", clip_weights="synthetic:0", clip_geometry=clipg, clip_text_geometry=dict(width=512, layers=12),
                     encoder=enc, vocab=voc, bpe_path=bpe, target_features=tf, pop_size=P, max_pop=P)
    prob = GenerationProblem(cfg)
    gen = prob.generator
    assert len(gen.model.init_tokens) == 3
    x = np.random.RandomState(11).randint(0, 50257, size=(P, cfg.dim_z))
    # ---- oracle chain ------------------------------------------------------------------------------------------------------------
    sdg = synth.make_state(synth.gpt2_spec(), 5)
    ctx = np.concatenate([x, np.tile(gen.model.init_tokens, (P, 1))], axis=1)
    dd = {}
    ora = gpt2_ref.sample_sequence(_t(sdg), torch.tensor(ctx), cfg.max_tokens_len, detail=dd).numpy()
    ref_texts = gpt2_ref.parse_out(ora, cfg.dim_z, gen.model.enc.eot, gen.model.enc.decode, cfg.max_text_len)
    sdt = synth.make_state(synth.clip_text_spec(width=512, layers=12, vocab=49408, out_dim=512), 0)
    tok = gen.tokenizer.tokenize(ref_texts)                 # (a failure here would zero the population, generator.py:53-56: the synthetic
    tfeat = clip_ref.encode_text(_t(sdt), torch.tensor(tok))  # vocabulary is built so that 50-character texts fit 77 tokens)
    # the target image feature: a direction the population's text features correlate with (a random one is orthogonal to all of them —
    # similarities ~0, where a relative bar means nothing): their normalised mean plus an equal-norm random part
    fn = torch.nn.functional.normalize(tfeat, dim=1).mean(0)
    tgt = (fn / fn.norm() + torch.nn.functional.normalize(torch.tensor(tf), dim=0)).numpy().astype(np.float32)
    gen.image_features = tgt[None]
    sim = torch.cosine_similarity(tfeat, torch.tensor(tgt)[None]).numpy()
    # ---- the product path ----------------------------------------------------------------------------------------------------------
    out = {}
    prob._evaluate(x, out)
    texts = list(gen.last_texts)
    assert out["F"].shape == (P,) and len(texts) == P
    got_tok = gen.engine.gpt2_decode(ctx, cfg.max_tokens_len)
    mg = dd["margins"].numpy() if hasattr(dd["margins"], "numpy") else dd["margins"]
    excepted = _assert_token_parity("C5 full size", got_tok, ora, mg, ctx.shape[1])
    assert all(a == b for i, (a, b) in enumerate(zip(texts, ref_texts)) if i not in excepted)
    F = np.asarray(out["F"])
    rel = np.abs(F + sim) / np.abs(sim)
    ok = np.ones(P, bool)
    ok[list(excepted)] = False
    diag("[gpt2] C5 full size: P=%d, max rel err of F vs oracle %.2e (sim %.3f..%.3f), %d near-tie row(s) excepted; sample %r"
         % (P, rel[ok].max(), sim.min(), sim.max(), len(excepted), texts[:2]))
    assert np.abs(sim).min() > 0.05                          # (the relative bar below means something)
    assert rel[ok].max() <= 1e-3
    gen.engine.close()


@pytest.mark.gpu
def test_gpt2_small_full_size_decode_and_timing():
    """GPT-2 small at true size (12 x 768, vocab 50257): P=64 x 23-token context, 30 greedy steps (config C5 shape);
    token parity of ALL 64 rows against the oracle + wall time of the device decode."""
    import time
    import glass_models as M
    from clip_glass_amd.engine import Engine
    sd = synth.make_state(synth.gpt2_spec(), 5)
    clip = M.CONFIGS["mini"]["clip"]
    sd.update(synth.make_state(synth.clip_visual_spec(clip[0], clip[1], clip[3], clip[4], clip[5]), 0))
    e = Engine([], latent_size=4, mapping_layers=0, batch_size=1, use_discriminator=False, n_obj=1, max_pop=64, clip=clip, noise_mode=0)
    e.load_state(sd)
    e.finalize()
    ctx = np.concatenate([_ctx(9, 64, 20, 50257), np.tile([1169, 4286, 286], (64, 1))], axis=1)
    e.gpt2_decode(ctx[:4], 2)                       # warm-up
    t = time.time()
    got = e.gpt2_decode(ctx, 30)
    dt = time.time() - t
    e.close()
    t = time.time()
    detail = {}
    ora = gpt2_ref.sample_sequence(_t(sd), torch.tensor(ctx), 30, detail=detail).numpy()
    dto = time.time() - t
    diag("[gpt2] GPT-2 small P=64 x 30 steps: device %.3f s (%.0f candidates/s); oracle 64 candidates %.2f s" % (dt, 64 / dt, dto))
    mg = detail["margins"].numpy() if hasattr(detail["margins"], "numpy") else detail["margins"]
    _assert_token_parity("GPT-2 small P=64", got, ora, mg, 23)


@pytest.mark.gpu
def test_encode_image_matches_oracle():
    import glass_models as M
    from oracle import clip_ref
    sd = M.make_state("mini", 0)
    e = M.make_engine("mini", sd, max_pop=8, noise_mode=0)
    img = synth.normal(4, "img", (3, 3, 32, 32), 1.0)
    got = e.encode_image(img)
    e.close()
    ref = clip_ref.encode_image(_t(sd), torch.tensor(img)).numpy()
    from util import check
    check("encode_image", got, ref, 5e-3)


def _gold():
    import os
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpt2_mini.npz")))


def test_oracle_reproduces_reference_gpt2_golden():
    g = _gold()
    sd = synth.make_state(synth.gpt2_spec(int(g["n_embd"]), int(g["n_layer"]), int(g["vocab"]), n_positions=64), int(g["seed"]))
    out = gpt2_ref.sample_sequence(_t(sd), torch.tensor(g["context"]), 30).numpy()
    assert np.array_equal(out, g["tokens"])
    with torch.no_grad():
        logits, _ = gpt2_ref.forward(_t(sd), torch.tensor(g["context"]))
    np.testing.assert_allclose(logits[:, -1, :64].numpy(), g["last_logits"], rtol=2e-4, atol=2e-5)


@pytest.mark.gpu
def test_engine_reproduces_reference_gpt2_golden():
    import glass_models as M
    from clip_glass_amd.engine import Engine
    g = _gold()
    sd = synth.make_state(synth.gpt2_spec(int(g["n_embd"]), int(g["n_layer"]), int(g["vocab"]), n_positions=64), int(g["seed"]))
    clip = M.CONFIGS["mini"]["clip"]
    sd.update(synth.make_state(synth.clip_visual_spec(clip[0], clip[1], clip[3], clip[4], clip[5]), 0))
    e = Engine([], latent_size=4, mapping_layers=0, batch_size=1, use_discriminator=False, n_obj=1, max_pop=8, clip=clip, noise_mode=0)
    e.load_state(sd)
    e.finalize()
    got = e.gpt2_decode(g["context"], 30)
    e.close()
    # the fixture holds the reference sampler's tokens; the oracle reproduces them exactly (test_oracle_reproduces_... on the CPU) and
    # supplies the per-step top-2 margins the near-tie escape needs
    dd = {}
    ora = gpt2_ref.sample_sequence(_t(sd), torch.tensor(g["context"]), 30, detail=dd).numpy()
    assert np.array_equal(ora, g["tokens"])
    mg = dd["margins"].numpy() if hasattr(dd["margins"], "numpy") else dd["margins"]
    _assert_token_parity("golden (reference sampler)", got, g["tokens"], mg, g["context"].shape[1])
