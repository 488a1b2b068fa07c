"""Host-side mirror of the reference's ``ponder`` package for the pre-training hot path:
registries, config loader, SpUNet-v1m1, PonderIndoor-v2, UNet3D-v1m2, the NeuS render head, a
DDP trainer.  Names registered here are the ones the reference's configs use."""
