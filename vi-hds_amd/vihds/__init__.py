"""MI355X-native implementation of vi-hds's batched ODE-integration + ELBO hot path, behind the reference's own
module names (vihds.ode / decoders / distributions / training / vae ..., models.LOOKUP)."""
