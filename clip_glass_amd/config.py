"""Config table — mirror of /root/reference/config.py (same keys and values for the
StyleGAN2 configs; `latent` / `model` point at this package's classes)."""
from .latent import DeepMindBigGANLatentSpace, GPT2LatentSpace, StyleGAN2LatentSpace
from .models import GPT2, DeepMindBigGAN, StyleGAN2
from .utils import biggan_denorm, biggan_norm


def _sg2(weights, use_d):
    return dict(
        task="txt2img", dim_z=512, latent=StyleGAN2LatentSpace, model=StyleGAN2, use_discriminator=use_d,
        weights=weights, algorithm="nsga2" if use_d else "ga", norm=biggan_norm, denorm=biggan_denorm,
        pop_size=16, batch_size=4,
        problem_args=dict(n_var=512, n_obj=2 if use_d else 1, n_constr=512, xl=-10, xu=10))


configs = dict(
    GPT2=dict(task="img2txt", dim_z=20, max_tokens_len=30, max_text_len=50, encoder_size=50257,
              latent=GPT2LatentSpace, model=GPT2, use_discriminator=False, init_text="the picture of",
              weights="./gpt2/weights/gpt2-pytorch_model.bin", encoder="./gpt2/weights/encoder.json",
              vocab="./gpt2/weights/vocab.bpe", stochastic=False, algorithm="ga", pop_size=100, batch_size=25,
              problem_args=dict(n_var=20, n_obj=1, n_constr=20, xl=0, xu=50256)),
    DeepMindBigGAN256=dict(task="txt2img", dim_z=128, num_classes=1000, latent=DeepMindBigGANLatentSpace,
                           model=DeepMindBigGAN, weights="biggan-deep-256", use_discriminator=False, algorithm="ga",
                           norm=biggan_norm, denorm=biggan_denorm, truncation=1.0, pop_size=64, batch_size=32,
                           problem_args=dict(n_var=128 + 1000, n_obj=1, n_constr=128, xl=-2, xu=2)),
    DeepMindBigGAN512=dict(task="txt2img", dim_z=128, num_classes=1000, latent=DeepMindBigGANLatentSpace,
                           model=DeepMindBigGAN, weights="biggan-deep-512", use_discriminator=False, algorithm="ga",
                           norm=biggan_norm, denorm=biggan_denorm, truncation=1.0, pop_size=32, batch_size=8,
                           problem_args=dict(n_var=128 + 1000, n_obj=1, n_constr=128, xl=-2, xu=2)),
    StyleGAN2_ffhq_d=_sg2("./stylegan2/weights/ffhq-config-f", True),
    StyleGAN2_car_d=_sg2("./stylegan2/weights/car-config-f", True),
    StyleGAN2_church_d=_sg2("./stylegan2/weights/church-config-f", True),
    StyleGAN2_ffhq_nod=_sg2("./stylegan2/weights/ffhq-config-f", False),
    StyleGAN2_car_nod=_sg2("./stylegan2/weights/car-config-f", False),
    StyleGAN2_church_nod=_sg2("./stylegan2/weights/church-config-f", False),
)


def get_config(name):
    return configs[name]
