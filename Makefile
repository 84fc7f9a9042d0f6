all:
	$(MAKE) -C clip_glass_amd/csrc

.PHONY: all
