/* embed.S -- the gfx950 code object (built by hipcc --genco from kernels.hip), carried inside
 * libhare_hip.so and loaded with hipModuleLoadData. */
    .section .rodata
    .balign 4096
    .global hare_kernels_co
    .global hare_kernels_co_end
hare_kernels_co:
    .incbin "hare_kernels.co"
hare_kernels_co_end:
    .byte 0
    .section .note.GNU-stack,"",@progbits
