"""MI355X-native RIED-GAN hot path (generator, discriminator, losses, metrics, training step)."""
