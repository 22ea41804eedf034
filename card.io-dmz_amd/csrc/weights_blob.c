/* weights_blob.c -- embeds card.io-dmz_amd/weights/dmz_models.bin (model data
 * extracted by tools/extract_models.py) into libdmz_hip.so, so that the shared
 * library is self-contained like the reference, whose generated model files
 * carry their weights as byte arrays (models/generated/ *.cpp). */
#ifndef DMZ_WEIGHTS_PATH
#error "DMZ_WEIGHTS_PATH must point at dmz_models.bin"
#endif
__asm__(
    ".section .rodata\n"
    ".balign 16\n"
    ".global dmz_weights_blob\n"
    "dmz_weights_blob:\n"
    ".incbin \"" DMZ_WEIGHTS_PATH "\"\n"
    ".global dmz_weights_blob_end\n"
    "dmz_weights_blob_end:\n"
    ".byte 0\n"
    ".previous\n");
