"""Model registry (filled in models/registry below once the plugin classes are defined)."""
