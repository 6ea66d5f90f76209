"""matpbr: MI355X-native differentiable PBR shading (hot path of Materialist)."""
